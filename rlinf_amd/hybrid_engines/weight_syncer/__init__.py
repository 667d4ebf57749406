from .patch_syncer import (  # noqa: F401
    EmptyWeightPatch,
    PatchBuilder,
    PatchWeightSyncer,
    WeightPatch,
    as_coo_2d_view,
    downscale_nonnegative_indices,
)
