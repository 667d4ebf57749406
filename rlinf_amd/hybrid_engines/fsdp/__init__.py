"""Host-side helpers of the FSDP learner path that sit on the hot path's data formats (sequence packing)."""
