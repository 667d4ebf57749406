"""Mirror of the slice of rlinf/hybrid_engines on the actor -> rollout weight path (weight_syncer)."""
