"""`cpp_wrappers` package with the reference's module paths:
`cpp_wrappers.cpp_subsampling.grid_subsampling.subsample_batch` and
`cpp_wrappers.cpp_neighbors.radius_neighbors.batch_query` (Predator_APR/datasets/dataloader.py:5-6)."""
