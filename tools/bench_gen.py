"""Generation leg of bench.py alone (BASELINE configs[3]): 32 streams, 64-token prompt -> 2048 tokens, nucleus p = 0.9.
Arguments: the EMO_DECODE_PERSISTENT settings to time, default "1 0" (one-launch step / chain of launches)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == '__main__':
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    torch.cuda.set_device(0)
    C = bench.CFG
    torch.manual_seed(0)
    model = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], dropout=0.1,
                           favor_feature_dims=C['n_feat'] if 'n_feat' in C else 128, use_segment_emb=True, n_segment_types=2, compute_dtype='bf16').cuda()
    for mode in (sys.argv[1:] or ['1', '0']):
        os.environ['EMO_DECODE_PERSISTENT'] = mode
        r = bench.generation_bench(model)
        r['persistent'] = mode
        print(json.dumps(r), flush=True)
