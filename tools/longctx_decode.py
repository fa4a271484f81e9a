#!/usr/bin/env python3
"""Whole-model greedy decode over LONG runs (one GPU, hipGraph token loop): the split attention launch (a head's keys over up
to 16 workgroups, csrc/decode_ops.hip) against one workgroup per head.   python tools/longctx_decode.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd.llama_decode import DecodeStage  # noqa: E402

dev = torch.device("cuda:0")
for ctx, toks in ((512, 448), (2048, 1900)):
    ids = {}
    for split in (False, True):
        st = DecodeStage(range(32), dev, max_ctx=ctx)
        if not split:
            st.attn_splits = 1
        tb = torch.zeros(1, dtype=torch.int64, device=dev)
        st.capture_token_loop(tb)
        st.reset(); st.decode_tokens(tb, 1, 8); st.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = st.decode_tokens(tb, 1, toks)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"KV cache {ctx}, {toks} tokens from position 0, {'split over <= ' + str(st.attn_splits) + ' workgroups' if split else 'one workgroup per head'}: "
              f"{toks / dt:.1f} tokens/s  {dt / toks * 1e3:.3f} ms/token", flush=True)
        ids[split] = out
        del st
    # the two launches sum a head's softmax in different orders (one pass vs merged parts): the ids agree until the first
    # argmax that a last-bit difference flips, and the runs are independent sequences from there on
    same = next((i for i, (a, b) in enumerate(zip(ids[False], ids[True])) if a != b), None)
    print(f"KV cache {ctx}: token ids of the two launches " + ("identical over all %d tokens" % toks if same is None
          else f"identical for the first {same} of {toks} tokens (first difference at position {same}: {ids[False][same]} vs {ids[True][same]})"),
          flush=True)
