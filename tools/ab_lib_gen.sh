#!/bin/bash
# same-box A/B of two builds of the library (emo-disentanger_amd/_ab_old.so vs _ab_new.so) on the two generation loops (Performer: tools/gen_prof.py,
# GPT-2 KV cache: tools/gen_prof_gpt2.py; 32 streams)
P=emo-disentanger_amd
export N_NEW=${N_NEW:-1984}
for v in old new old new; do
  cp $P/_ab_$v.so $P/libemo_hip.so
  echo "$v performer: $(python tools/gen_prof.py 2>/dev/null | tail -1)"
  echo "$v gpt2:      $(python tools/gen_prof_gpt2.py 2>/dev/null | tail -1)"
done
