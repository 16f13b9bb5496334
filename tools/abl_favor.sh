for b in 0 1 4 8 16 32 2 64 128 255; do echo -n "dkv abl=$b "; BS=64 EMO_FAVOR_ABLATE_DKV=$b python tools/bench_favor.py; done
for b in 1 2 4 8 16 32 64 128 255; do echo -n "dq abl=$b "; BS=64 EMO_FAVOR_ABLATE_DQ=$b python tools/bench_favor.py; done
