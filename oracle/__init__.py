"""ORACLE — test infrastructure, NOT product code.

CPU restatement (PyTorch fp32 eager ops + one plain-C kernel) of the reference's
stage-2 hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``emo-disentanger_amd/``) never does.

Pinning status (see DESIGN.md "Oracle"):
  * prologue (embedding + segment embedding + PE), GPT-2 block stack, logits,
    cross-entropy, accuracy, LR schedule, temperature / nucleus sampling and the
    ``generate_conditional`` control flow are PINNED: ``tools/make_golden.py``
    imports the real reference from /root/reference in the build container and
    the resulting vectors live in ``tests/golden/`` (tests/test_oracle_golden.py).
  * the Performer attention arithmetic (FAVOR+ feature map, causal linear
    attention, post-LN encoder layer) lives in the un-vendored, un-pinned
    third-party ``pytorch-fast-transformers`` => **parity unpinned** for those
    functions; they are restated from the published algorithm and checked by
    independent identities (tests/test_oracle_performer.py).
"""
