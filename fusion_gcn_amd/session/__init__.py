"""Host-side mirror of the reference's training harness around the hot path (torch_src/session/: SURVEY.md section 8 rows a10 /
a11): the per-batch step objects, the batch processors and the epoch loops, with the HIP-graph step as the MI355X-native one."""
