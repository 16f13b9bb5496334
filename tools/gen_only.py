import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd.model.music_performer import MusicPerformer
C = bench.CFG
torch.manual_seed(0)
m = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], favor_feature_dims=C['n_feat'],
                   use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda()
print(json.dumps(bench.generation_bench(m, n_new=int(os.environ.get('NNEW', 128)))))
