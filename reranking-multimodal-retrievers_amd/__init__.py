"""rmr_amd — MI355X-native cross-encoder rerank forward behind the reference's reranker plug-in API.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of librerank_mi355.so),
`_lib.py` (ctypes declarations), `model.py` (mirror of the reference's RerankerClass interface),
`pair_inputs.py` (pair-input assembly), `sharding.py` (pair sharding + score all-gather across ranks), `evaluate.py` (batched executor-side rerank loop, the
reference's prediction-record schema and Recall@K).
Importing the package does not need a GPU; constructing a model does (no CPU fallback exists).
"""
from ._lib import EXPORTED, LIB_PATH  # noqa: F401
from .model import (FullContextRerankModel, InteractionRerankModel, RerankModel, RerankEngine, RerankOutput, make_arch,  # noqa: F401
                    synthetic_state_dict, weight_spec)
from .sharding import shard_range, ShardedReranker  # noqa: F401
from .ranking import rank_descending_stable, recall_precision_at_k  # noqa: F401
from .evaluate import build_records, compute_rerank_scores, rerank_dataset  # noqa: F401

__version__ = "0.1.0"
