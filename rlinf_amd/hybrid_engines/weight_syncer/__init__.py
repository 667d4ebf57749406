from .base import WeightSyncer  # noqa: F401
from .bucket_syncer import (  # noqa: F401
    BucketWeightSyncer,
    WeightBucket,
    iter_named_tensor_buckets,
    load_bucket,
    plan_buckets,
)
from .compressor import IdentityCompressor, PatchCompressor, ZPlaneCompressor  # noqa: F401
from .patch_syncer import (  # noqa: F401
    CompressedWeightPatch,
    CPUSnapshotPatchBuilder,
    EmptyWeightPatch,
    PatchBuilder,
    PatchWeightSyncer,
    WeightPatch,
    as_coo_2d_view,
    downscale_nonnegative_indices,
)
