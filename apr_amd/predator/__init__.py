"""Host-side mirror of the Predator_APR hot path: grid subsampling, radius neighbours, collate,
KPConv encoder + overlap attention, score sampling + RANSAC.  Layout follows
/root/reference/Predator_APR (cpp_wrappers, datasets/dataloader.py, models/, lib/tester.py)."""
