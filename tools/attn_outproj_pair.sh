#!/bin/bash
# Verdict r5 item 4: the attention-output store policy decided on the PAIR (attention + out-proj, in-tower times), and out-proj's activation DMA with
# the matching load policy.  attn_loader 2 = non-temporal output rows (product), 1 = default-policy stores; libclipmi_rsaux{1,2,3}.so = the row-range
# out-proj kernel with cache-policy bits 1 / 2 / 3 on the LDS-DMA of its activation operand (build: see profiles/r06_attn_outproj_pair.txt).
cd "$(dirname "$0")/.."
for al in 2 1; do
  echo "== attn_loader=$al"
  OPTIONS=attn_loader=$al ROUNDS=3 python tools/lib_tower_ab.py libclipmi.so libclipmi_rsaux1.so libclipmi_rsaux2.so libclipmi_rsaux3.so
done
