"""Sparse-training seam of the hot path (RigL subset): per-step mask application is fused into the
engine's Adam kernel; topology updates (ERK init, magnitude prune, |grad| growth, cosine decay) are
index paths reproduced bit-exactly from the reference (implicit_image/pipeline/masking/)."""
from .core import LayerStats, Masking
from .funcs import (CosineDecay, LinearDecay, MagnitudePruneDecay, decay_registry, erdos_renyi_densities, grow_registry, init_registry,
                    prune_registry, redistribute_registry)

__all__ = ["Masking", "LayerStats", "CosineDecay", "LinearDecay", "MagnitudePruneDecay", "decay_registry", "erdos_renyi_densities", "grow_registry",
           "init_registry", "prune_registry", "redistribute_registry"]
