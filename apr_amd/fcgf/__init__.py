"""Host-side mirror of the FCGF_APR hot path (encoder, feature matching, pose fit).

Module layout and call signatures follow /root/reference/FCGF_APR so the
reference's train/eval scripts can import these in place of their own
`model`, `lib.eval`, `lib.metrics` and `util.transform_estimation` modules.
"""
