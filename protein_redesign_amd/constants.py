"""Vocabulary sizes and defaults that size the hot path's tables.

Only the *cardinalities* of the reference's chemistry vocabularies matter here
(reference ProteinReDiff/features.py:31-60, ProteinReDiff/protein.py:28-31);
rdkit / Biopython are not needed on the hot path.
"""

# categorical atom feature table sizes, in the order of features.py:31-46
ATOM_FEATURE_CARDS = (119, 4, 12, 12, 10, 6, 6, 2, 2)
# categorical bond feature table sizes, features.py:49-60
BOND_FEATURE_CARDS = (5, 6, 2)
# 20 amino acids (protein.py:28-31); logits have one extra "X"/pad class in front
RESIDUE_TYPES = [
    "A", "R", "N", "D", "C", "Q", "E", "G", "H", "I",
    "L", "K", "M", "F", "P", "S", "T", "W", "Y", "V",
]
NUM_RESIDUE_CLASSES = len(RESIDUE_TYPES) + 1  # 21
NUM_RESIDUE_ATOMS = 37                        # protein.py:41-47 (index 1 = CA)

# ProteinReDiffModel.add_argparse_args defaults (model.py:137-170)
DEFAULT_ARGS = dict(
    training_mode=False, mask_prob=1.0, esm_dim=1280, time_dim=256, dist_dim=256,
    single_dim=512, pair_dim=64, head_dim=16, num_heads=4, transition_factor=4,
    num_blocks=12, max_bond_distance=7, max_relpos=32, num_steps=64,
    diffusion_schedule="linear", learning_rate=4e-4, warmup_steps=1000,
    ema_decay=0.999, n_recycles=4, top_k_neighbors=30, dropout=0.3,
    num_gvp_encoder_layers=3, num_positional_embeddings=16,
    gvp_edge_hidden_dim_scalar=32, gvp_edge_hidden_dim_vector=32,
)


def make_args(**overrides):
    """Namespace-like dict of model hyper-parameters (model.py:56-77)."""
    args = dict(DEFAULT_ARGS)
    args.update(overrides)
    return args
