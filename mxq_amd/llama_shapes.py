"""Llama-2-7B Linear shapes (reference LLM-QAT/models/configuration_llama.py:84-88,
modeling_llama_quant.py:210-230,262-291): hidden 4096, intermediate 11008, 32 layers,
7 Linears per layer; 202,375,168 Linear parameters per layer, 6,476,005,376 in total
(lm_head is a plain nn.Linear and is not quantised, modeling_llama_quant.py:795)."""
from __future__ import annotations

HIDDEN = 4096
INTERMEDIATE = 11008
N_LAYERS = 32

# name -> (out_features N, in_features K)
LAYER_LINEARS = (
    ("q_proj", HIDDEN, HIDDEN),
    ("k_proj", HIDDEN, HIDDEN),
    ("v_proj", HIDDEN, HIDDEN),
    ("o_proj", HIDDEN, HIDDEN),
    ("gate_proj", INTERMEDIATE, HIDDEN),
    ("up_proj", INTERMEDIATE, HIDDEN),
    ("down_proj", HIDDEN, INTERMEDIATE),
)

PARAMS_PER_LAYER = sum(n * k for _, n, k in LAYER_LINEARS)
assert PARAMS_PER_LAYER == 202_375_168


def layer_range(rank: int, world: int, n_layers: int = N_LAYERS):
    """Whole-layer sharding of SURVEY.md 8e: rank r of G owns layers [32r/G, 32(r+1)/G)."""
    return range(n_layers * rank // world, n_layers * (rank + 1) // world)


def linear_flops(M: int, layers: int = N_LAYERS) -> float:
    return 2.0 * M * PARAMS_PER_LAYER * layers
