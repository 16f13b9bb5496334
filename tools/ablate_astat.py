import os, sys, subprocess
for ab in ('0', '1', '2', '3'):
    env = dict(os.environ, EMO_GEMM_ABLATE=ab)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'bench_astat.py')], env=env, capture_output=True, text=True)
    print('ablate', ab)
    print('\n'.join(l for l in r.stdout.split('\n') if 'QKV' in l or 'plain' in l or 'FFN1 fwd (bias' in l))
