#!/bin/bash
# same-box A/B of one environment setting in the GPT-2 KV-cache generation (tools/gen_prof_gpt2.py, 32 streams, 64-token prompt + N_NEW tokens):
# tools/ab_gen_gpt2.sh NAME VALUE_A VALUE_B  (VALUE "-" = unset)
N=$1; A=$2; B=$3
export N_NEW=${N_NEW:-1984}
for v in "$A" "$B" "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $N; else export $N="$v"; fi
  echo "$N=$v $(python tools/gen_prof_gpt2.py 2>/dev/null | tail -1)"
done
