#!/bin/bash
# same-box A/B of two builds of the library (emo-disentanger_amd/_ab_old.so vs _ab_new.so) on the GPT-2 backbone's training step (tools/bench_gpt2.py)
P=emo-disentanger_amd
for v in old new old new; do
  cp $P/_ab_$v.so $P/libemo_hip.so
  echo "$v $(python tools/bench_gpt2.py 2>/dev/null | tail -1 | cut -c1-400)"
done
