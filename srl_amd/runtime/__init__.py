"""Host-side runtime pieces either side of the hot path: synthetic sample generator, inference
batcher, sample buffer, data-parallel helpers."""
