#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN
PYTHON on CPU (read-only tree at /root/reference; it never ships to the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only data is written (inputs and the reference's outputs, as .npz; bf16 tensors as
uint16 bit patterns).  Nothing from the reference's source text is stored.

Reference entry points exercised:
  G1 ptq_small   lib.mxqgpt.MXQGPT.add_batch/fasterquant  +  lib.quantizer.Quantizer
                 called group by group exactly as fasterquant does (mxqgpt.py:417-436)
                 to expose the integer codes / parameters it keeps in locals.
  G2 ptq_slices  the same on Llama-shaped [1024,4096] / [256,11008]: SHA-256 of every
                 tensor + one 16-row slice in full.
  G3 qat_small   models.utils_quant.MXAsymQuantizer fwd/bwd in fp32 / bf16 / fp16.
  G4 qlinear     models.utils_quant.QuantizeLinear fwd/bwd (fp32, bf16).
  G5 block_small models.modeling_llama_quant.LlamaDecoderLayer fwd/bwd (w_bits=2).
  G6 kat         constants of cuda_kernel/test_correct_gemv.py (expected == 4096).
  G7 act         models.utils_quant.SymQuantizer / AsymQuantizer fwd/bwd (2-D, 3-D, 4-D).
  G8 uniform     lib.quantizer.Quantizer as uniform W2 (group 16) / W4 (per row) quantiser.
  G9 act16       SymQuantizer / AsymQuantizer forward on bf16 / fp16 tensors (uint16 bit patterns).

    python tests/golden/make_golden.py g9        # regenerate selected sets only
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("MXQ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "mxq_quant"))
sys.path.insert(0, os.path.join(REF, "LLM-QAT"))

# MXQGPT.fasterquant/free call these unconditionally (mxqgpt.py:445,452); on a CPU-only
# box they raise an ordinary RuntimeError, so they are stubbed (reference files untouched).
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None

import lib.mxqgpt as ref_mxqgpt          # noqa: E402
import lib.quantizer as ref_quantizer    # noqa: E402
from models.utils_quant import AsymQuantizer, MXAsymQuantizer, QuantizeLinear, SymQuantizer   # noqa: E402


def bf16_bits(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_ptq(W16, x_calib):
    """Run the reference PTQ on an fp16 weight; return codes/params/W_deq."""
    N, K = W16.shape
    lin = nn.Linear(K, N, bias=False)
    lin.weight.data = W16.clone()
    g = ref_mxqgpt.MXQGPT(lin)
    g.add_batch(x_calib, None)
    dead = (torch.diag(g.H) == 0).numpy().copy()
    g.fasterquant(percdamp=0.01, blocksize=16)          # prune.py:409
    W_deq = lin.weight.data.clone()                     # fp16 (mxqgpt.py:448)
    g.free()

    # codes and parameters: same Quantizer calls, same order, same operands as
    # fasterquant (mxqgpt.py:417-436).
    W = W16.float().clone()
    W[:, torch.from_numpy(dead)] = 0
    nc = K // 64
    codes2 = torch.zeros(N, nc * 3, 16)
    sc2 = torch.zeros(N, nc * 3)
    zero2 = torch.zeros(N, nc * 3)
    scale2 = torch.zeros(N, nc * 3)
    qs2 = torch.zeros(N // 16, nc * 3)
    qz2 = torch.zeros(N // 16, nc * 3)
    W4 = torch.zeros(N, nc * 16)
    Wchk = W.clone()
    for c in range(nc):
        for gi in range(3):
            lo = c * 64 + gi * 16
            W1 = W[:, lo:lo + 16].clone()
            q = ref_quantizer.Quantizer()
            q.configure(bits=2, perchannel=True, sym=False, qq_scale_bits=4)
            q.find_params(W1, weight=True)
            j = 3 * c + gi
            codes2[:, j] = q.quantize(W1)
            sc2[:, j] = q.quant_scale.reshape(-1)
            zero2[:, j] = q.zero.reshape(-1)
            scale2[:, j] = q.scale.reshape(-1)
            qs2[:, j] = q.qq_scale.scale.reshape(-1)
            qz2[:, j] = q.qq_scale.zero.reshape(-1)
            Wchk[:, lo:lo + 16] = q.quantize_dequantize(W1)
        W4[:, c * 16:(c + 1) * 16] = W[:, c * 64 + 48:(c + 1) * 64]
    q4 = ref_quantizer.Quantizer()
    q4.configure(bits=4, perchannel=True, sym=False, qq_scale_bits=4)
    q4.find_params(W4, weight=True)
    codes4 = q4.quantize(W4)
    W4q = q4.quantize_dequantize(W4)
    for c in range(nc):
        Wchk[:, c * 64 + 48:(c + 1) * 64] = W4q[:, c * 16:(c + 1) * 16]
    # the group-by-group replay must reproduce fasterquant's own output bit for bit
    assert torch.equal(Wchk.half(), W_deq), "replay of fasterquant diverged"
    return dict(
        dead=dead,
        codes2=codes2.reshape(N, nc * 48).numpy().astype(np.uint8),
        sc2=sc2.numpy().astype(np.uint8), zero2=zero2.numpy(), scale2=scale2.numpy(),
        qs2=qs2.numpy(), qz2=qz2.numpy(),
        codes4=codes4.numpy().astype(np.uint8),
        sc4=q4.quant_scale.reshape(-1).numpy().astype(np.uint8),
        zero4=q4.zero.reshape(-1).numpy(), scale4=q4.scale.reshape(-1).numpy(),
        qs4=q4.qq_scale.scale.reshape(-1).numpy(), qz4=q4.qq_scale.zero.reshape(-1).numpy(),
        w_deq=W_deq.numpy(),
    )


def g1_ptq_small():
    torch.manual_seed(0)
    N, K = 64, 256
    W = torch.randn(N, K) * 0.02
    W[3, 16:32] = 0.0123          # constant group -> xmin == xmax branch (quantizer.py:90-92)
    W[5, 0:16] = -W[5, 0:16].abs() - 0.01     # all-negative group (zero-point > maxq)
    W[7, 64:80] = W[7, 64:80].abs() + 0.01    # all-positive group (negative zero-point)
    W[9, 40] = 0.9                # outliers
    W[9, 130] = -1.1
    W[11, 48:64] = 0.05           # constant 4-bit slice inside one chunk
    W[16:32, 100] *= 30.0         # stretches the second-order scale range of one row block
    W[40, :] = 0.25               # a whole constant row (4-bit arm xmin == xmax)
    W16 = W.half()
    x = torch.randn(8, K).half()
    x[:, 77] = 0                  # dead column: diag(H) == 0 (mxqgpt.py:401-403)
    r = ref_ptq(W16, x)
    assert r["dead"][77] and r["dead"].sum() == 1
    y32 = x.float() @ torch.from_numpy(r["w_deq"]).float().t()
    np.savez_compressed(os.path.join(OUT, "g1_ptq_small.npz"), W=W16.numpy(), x=x.numpy(),
                        y32=y32.numpy(), y16=y32.half().numpy(), **r)


def g2_ptq_slices():
    out = {}
    for name, (N, K, seed) in {"a": (1024, 4096, 1), "b": (256, 11008, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        W16 = (torch.randn(N, K, generator=g) * 0.02).half()
        x = torch.randn(4, K, generator=g).half()
        r = ref_ptq(W16, x)
        out[f"{name}_shape"] = np.array([N, K, seed])
        for k, v in r.items():
            out[f"{name}_sha_{k}"] = np.array(sha(v))
            if k != "dead":
                sl = v[16:32] if v.shape[0] == N else v[1:2]
                out[f"{name}_rows16_32_{k}"] = sl
    np.savez_compressed(os.path.join(OUT, "g2_ptq_slices.npz"), **out)


def qat_input(seed, N=64, K=256):
    torch.manual_seed(seed)
    w = torch.randn(N, K) * 0.05
    w[2, 5] = 2.0                 # exact clip boundaries (ge / le in backward)
    w[2, 6] = -2.0
    w[4, 70] = 2.5
    w[4, 71] = -3.0
    w[6, 0:16] = 0.03125          # constant 2-bit group (alpha == 0)
    w[8, 16:32] = -w[8, 16:32].abs()
    w[10, 48:64] = 1.5            # large 4-bit slice -> wide row range
    return w


def g3_qat_small():
    out = {}
    clip = torch.tensor([-2.0, 2.0])
    for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for bits in (2, 3, 4):
            w = qat_input(3).to(dt).requires_grad_()
            o = MXAsymQuantizer.apply(w, clip, bits, False)
            torch.manual_seed(11)
            go = torch.randn(w.shape).to(dt)
            o.backward(go)
            key = f"{name}_b{bits}"
            if dt == torch.bfloat16:
                out[f"{key}_w"], out[f"{key}_out"] = bf16_bits(w), bf16_bits(o)
                out[f"{key}_gout"], out[f"{key}_gin"] = bf16_bits(go), bf16_bits(w.grad)
            else:
                out[f"{key}_w"], out[f"{key}_out"] = w.detach().numpy(), o.detach().numpy()
                out[f"{key}_gout"], out[f"{key}_gin"] = go.numpy(), w.grad.numpy()
    # a Llama-width row set in bf16 (K = 4096 and 11008 / 4 = 2752 gathered 4-bit values)
    for K in (4096, 11008):
        torch.manual_seed(K)
        w = (torch.randn(16, K) * 0.02).bfloat16()
        o = MXAsymQuantizer.apply(w, clip, 2, False)
        out[f"bf16_K{K}_w"], out[f"bf16_K{K}_out"] = bf16_bits(w), bf16_bits(o)
    np.savez_compressed(os.path.join(OUT, "g3_qat_small.npz"), **out)


def g4_qlinear():
    out = {}
    for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        torch.manual_seed(5)
        lin = QuantizeLinear(256, 64, w_bits=2, a_bits=16)
        lin.weight.data = (torch.randn(64, 256) * 0.05)
        lin = lin.to(dt)
        x = torch.randn(2, 8, 256).to(dt).requires_grad_()
        y = lin(x)
        go = torch.randn(y.shape).to(dt)
        y.backward(go)
        conv = bf16_bits if dt == torch.bfloat16 else (lambda t: t.detach().numpy())
        out[f"{name}_w"], out[f"{name}_x"] = conv(lin.weight), conv(x)
        out[f"{name}_y"], out[f"{name}_gy"] = conv(y), conv(go)
        out[f"{name}_dx"], out[f"{name}_dw"] = conv(x.grad), conv(lin.weight.grad)
    np.savez_compressed(os.path.join(OUT, "g4_qlinear.npz"), **out)


def g5_block_small():
    from models.configuration_llama import LlamaConfig
    from models.modeling_llama_quant import LlamaDecoderLayer
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_attention_heads=4,
                      num_hidden_layers=1, vocab_size=128, max_position_embeddings=64)
    cfg.w_bits, cfg.a_bits, cfg.kv_bits = 2, 16, 16      # train.py:56-58
    torch.manual_seed(21)
    layer = LlamaDecoderLayer(cfg)
    x = (torch.randn(2, 16, 256) * 0.5).requires_grad_()
    pos = torch.arange(16).unsqueeze(0).expand(2, -1)
    mask = torch.full((16, 16), torch.finfo(torch.float32).min).triu(1)[None, None].expand(2, 1, 16, 16)
    y = layer(x, attention_mask=mask, position_ids=pos)[0]
    torch.manual_seed(22)
    go = torch.randn(y.shape)
    y.backward(go)
    out = dict(x=x.detach().numpy(), y=y.detach().numpy(), gy=go.numpy(), dx=x.grad.numpy(),
               pos=pos.numpy())
    for k, v in layer.state_dict().items():
        out["sd_" + k] = v.numpy()
    for k, p in layer.named_parameters():
        out["grad_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_block_small.npz"), **out)


def g6_kat():
    # scalar constants of cuda_kernel/test_correct_gemv.py:19-37; expected output :51
    np.savez_compressed(
        os.path.join(OUT, "g6_kat.npz"),
        M=1, N=4096, K=4096, group_size=16,
        zeros_and_scales_1st=np.uint32(0xAA55AA55), zeros_2nd=np.uint32(0x55555555),
        scales_2nd=np.float16(1), weight_2b=np.uint32(0xAAAAAAAA), weight_4b=np.uint32(0xAAAAAAAA),
        zeros_4b=np.uint32(0x99999999), scales_4b=np.float16(1), x=np.float16(1),
        expected=np.int32(4096),
        shape_weight=np.array([4096, 256]), shape_weight_last=np.array([4096, 64]),
        shape_zeros_and_scales=np.array([4096, 32]), shape_zeros_2nd=np.array([1024, 32]),
        shape_scales_2nd=np.array([1024, 256]), shape_scales_4b=np.array([4096]),
        shape_zeros_4b=np.array([512]))


def g7_act_quantizers():
    """SymQuantizer / AsymQuantizer (utils_quant.py:31-199): outside the hot path, but
    QuantizeLinear(a_bits=16) and the decoder layer call SymQuantizer on activations."""
    out = {}
    clip = torch.tensor([-2.0, 2.0])
    cases = {"w2d": (8, 256), "a3d": (2, 8, 256), "a3d_long": (1, 140, 128), "s4d": (1, 2, 4, 4)}
    for cname, shape in cases.items():
        for qname, Q in (("sym", SymQuantizer), ("asym", AsymQuantizer)):
            for bits in (4, 16):
                for layerwise in (False, True):
                    torch.manual_seed(100 * len(cname) + bits)
                    x = (torch.randn(*shape) * 1.2).requires_grad_()
                    y = Q.apply(x, clip, bits, layerwise)
                    go = torch.randn(*shape)
                    y.backward(go)
                    key = f"{cname}_{qname}_b{bits}_{int(layerwise)}"
                    out[key + "_x"], out[key + "_y"] = x.detach().numpy(), y.detach().numpy()
                    out[key + "_gy"], out[key + "_gx"] = go.numpy(), x.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g7_act_quantizers.npz"), **out)


def g8_uniform():
    """Uniform W2 (group 16) / W4 (per row) arms of the config-5 sweep: the reference's Quantizer
    (lib/quantizer.py) applied group by group / to whole rows."""
    torch.manual_seed(8)
    N, K = 32, 128
    W = (torch.randn(N, K) * 0.02).half()
    W[1, 16:32] = 0.01          # constant group
    W[2, :] = -0.3              # constant row
    Wf = W.float()
    out = dict(W=W.numpy())
    codes = torch.zeros(N, K); sc = torch.zeros(N, K // 16); zero = torch.zeros(N, K // 16)
    qs = torch.zeros(N // 16, K // 16); qz = torch.zeros(N // 16, K // 16); wq = torch.zeros(N, K)
    for g in range(K // 16):
        blk = Wf[:, 16 * g:16 * g + 16].clone()
        q = ref_quantizer.Quantizer()
        q.configure(bits=2, perchannel=True, sym=False, qq_scale_bits=4)
        q.find_params(blk, weight=True)
        codes[:, 16 * g:16 * g + 16] = q.quantize(blk)
        wq[:, 16 * g:16 * g + 16] = q.quantize_dequantize(blk)
        sc[:, g], zero[:, g] = q.quant_scale.reshape(-1), q.zero.reshape(-1)
        qs[:, g], qz[:, g] = q.qq_scale.scale.reshape(-1), q.qq_scale.zero.reshape(-1)
    out.update(w2_codes=codes.numpy().astype(np.uint8), w2_sc=sc.numpy().astype(np.uint8), w2_zero=zero.numpy(),
               w2_qs=qs.numpy(), w2_qz=qz.numpy(), w2_wdeq=wq.half().numpy())
    q = ref_quantizer.Quantizer()
    q.configure(bits=4, perchannel=True, sym=False, qq_scale_bits=4)
    q.find_params(Wf.clone(), weight=True)
    out.update(w4_codes=q.quantize(Wf).numpy().astype(np.uint8),
               w4_sc=q.quant_scale.reshape(-1, 1).numpy().astype(np.uint8), w4_zero=q.zero.reshape(-1, 1).numpy(),
               w4_qs=q.qq_scale.scale.reshape(-1, 1).numpy(), w4_qz=q.qq_scale.zero.reshape(-1, 1).numpy(),
               w4_wdeq=q.quantize_dequantize(Wf).half().numpy())
    np.savez_compressed(os.path.join(OUT, "g8_uniform.npz"), **out)


def g9_act_quantizers_16bit():
    """SymQuantizer / AsymQuantizer forward on 16-bit tensors: PyTorch evaluates every op in fp32 and
    rounds to the tensor dtype, which the HIP kernels reproduce op by op."""
    out = {}
    clip = torch.tensor([-2.0, 2.0])
    cases = {"w2d": (16, 384), "w2d_ragged": (4, 200), "a3d": (2, 8, 256), "a3d_long": (1, 140, 128), "s4d": (1, 3, 4, 8)}
    for dname, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        for cname, shape in cases.items():
            for qname, Q in (("sym", SymQuantizer), ("asym", AsymQuantizer)):
                for bits in (4, 8, 16):
                    for layerwise in (False, True):
                        torch.manual_seed(7 * len(cname) + bits + len(dname))
                        x = (torch.randn(*shape) * 1.2).to(dt)
                        if cname == "w2d":
                            x[3, 128:256] = 0.5        # a constant group (asym: alpha == 0)
                            x[5, :128] = 0             # an all-zero group (sym: max == 0)
                        y = Q.apply(x, clip, bits, layerwise)
                        assert y.dtype == dt
                        key = f"{dname}_{cname}_{qname}_b{bits}_{int(layerwise)}"
                        out[key + "_x"], out[key + "_y"] = bf16_bits(x), bf16_bits(y)
    np.savez_compressed(os.path.join(OUT, "g9_act16.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    gens = [("g1", g1_ptq_small), ("g2", g2_ptq_slices), ("g3", g3_qat_small), ("g4", g4_qlinear),
            ("g5", g5_block_small), ("g6", g6_kat), ("g7", g7_act_quantizers), ("g8", g8_uniform),
            ("g9", g9_act_quantizers_16bit)]
    want = set(a.lower() for a in sys.argv[1:])
    for name, fn in gens:
        if not want or name in want:
            fn()
            print(name.upper(), "ok")
