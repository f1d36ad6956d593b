"""GPU collate for the KPConv encoder: per-layer points / neighbours / pools / upsamples.

Mirrors /root/reference/Predator_APR/datasets/dataloader.py: `batch_grid_subsampling_kpconv`
(:15-52), `batch_neighbors_kpconv` (:54-70), `collate_fn_descriptor` (:72-198) and
`calibrate_neighbors` (:200-232), with the same arguments and the same dict keys, but every
index build is a HIP kernel on device tensors (no DataLoader workers, no host round trips for
the point data).  Index tensors are int32 (the reference converts to int64; the blocks accept
both).
"""
import numpy as np
import torch

from .. import point_ops


def batch_grid_subsampling_kpconv(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                                  random_grid_orient=True):
    if labels is not None:
        raise NotImplementedError("labels are not used by the APR pipeline")
    res = point_ops.grid_subsample(points, batches_len, sampleDl, features)
    out = (res[0], torch.from_numpy(res[1].astype(np.int32)))
    return out + ((res[2],) if features is not None else ())


def batch_neighbors_kpconv(queries, supports, q_batches, s_batches, radius, max_neighbors):
    return point_ops.radius_neighbors(queries, supports, np.asarray(q_batches), np.asarray(s_batches), radius,
                                      limit=int(max_neighbors) if max_neighbors > 0 else 0)


def collate_fn_descriptor(list_data, config, neighborhood_limits):
    """The reference's collate (dataloader.py:72-198) on device tensors; blocking form of `collate_phases`."""
    from ... import ops
    return ops.drive(collate_phases(list_data, config, neighborhood_limits))


def collate_phases(list_data, config, neighborhood_limits):
    """list_data: [(src_pcd, tgt_pcd, src_feats, tgt_feats, ...extras)] with numpy or tensor clouds.

    The reference collates exactly one pair (dataloader.py:73 asserts it).  Several pairs may be stacked here
    ([src0, tgt0, src1, tgt1, ...] at every level): neighbourhoods never cross clouds, and the per-level row offsets of
    the pairs are recorded under 'pair_rows' so that the network keeps its per-pair normalisation statistics and runs
    its overlap attention pair by pair -- every pair gets the result of its own single-pair batch."""
    # A generator: it yields an ops.PendingFetch wherever the host needs bytes back from the device (the cloud lengths
    # of every pooled level, the table widths) and returns the batch dict -- a single-thread scheduler resumes it when
    # the fetch has landed (PredatorRegistration.register_batch_phases), `collate_fn_descriptor` just waits.
    assert len(list_data) >= 1
    item = list_data[0]
    dev = torch.device('cuda', torch.cuda.current_device())

    def to_dev(a):
        a = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        return a.to(device=dev, dtype=torch.float32)

    batched_points = torch.cat([to_dev(c) for it in list_data for c in it[:2]], 0).contiguous()
    batched_features = torch.cat([to_dev(f) for it in list_data for f in it[2:4]], 0).contiguous()
    batched_lengths = torch.tensor([len(c) for it in list_data for c in it[:2]], dtype=torch.int32)

    r_normal = config.first_subsampling_dl * config.conv_radius
    layer_blocks, layer = [], 0
    input_points, input_neighbors, input_pools, input_upsamples, input_batches_len = [], [], [], [], []
    empty_i = torch.zeros((0, 1), dtype=torch.int32, device=dev)
    # With calibrated limits the neighbour tables are built without host synchronisations (the reference's table width
    # min(max count, limit) is only needed to drop all-padding columns): their (max count, overflow) words are read
    # in ONE copy after the loop.  Same tensors as `batch_neighbors_kpconv` returns, 1 sync instead of 2 per table.
    deferred, slots = [], []
    flags_all = torch.empty((16, 2), dtype=torch.int32, device=dev)

    level_grid = point_ops.SearchGrid()     # the level's own points bucketed at the conv radius: searched twice

    def neighbors(queries, supports, q_b, s_b, radius, limit, where):
        if limit is None or int(limit) <= 0 or len(deferred) >= flags_all.shape[0]:
            return batch_neighbors_kpconv(queries, supports, q_b, s_b, radius, limit)
        t = point_ops.radius_neighbors_async(queries, supports, np.asarray(q_b), np.asarray(s_b), radius, int(limit),
                                             flags_all[len(deferred)], keep_grid=level_grid, grid=level_grid)
        deferred.append(t)
        slots.append(where)
        return t

    for block_i, block in enumerate(config.architecture):
        if 'global' in block or 'upsample' in block:
            break
        if not ('pool' in block or 'strided' in block):
            layer_blocks += [block]
            if block_i < len(config.architecture) - 1 and not ('upsample' in config.architecture[block_i + 1]):
                continue
        if layer_blocks:
            r = r_normal * config.deform_radius / config.conv_radius \
                if np.any(['deformable' in blck for blck in layer_blocks[:-1]]) else r_normal
            conv_i = neighbors(batched_points, batched_points, batched_lengths, batched_lengths, r,
                               neighborhood_limits[layer], (input_neighbors, len(input_neighbors)))
        else:
            conv_i = empty_i
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / config.conv_radius
            sub = point_ops.grid_subsample_async(batched_points, batched_lengths, dl)
            yield sub.fetch
            pool_p, pool_b = sub.finish()
            pool_b = torch.from_numpy(pool_b.astype(np.int32))
            r = r_normal * config.deform_radius / config.conv_radius if 'deformable' in block else r_normal
            pool_i = neighbors(pool_p, batched_points, pool_b, batched_lengths, r, neighborhood_limits[layer],
                               (input_pools, len(input_pools)))
            up_i = neighbors(batched_points, pool_p, batched_lengths, pool_b, 2 * r, neighborhood_limits[layer],
                             (input_upsamples, len(input_upsamples)))
        else:
            pool_i, up_i = empty_i, empty_i
            pool_p = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            pool_b = torch.zeros((0,), dtype=torch.int32)
        input_points += [batched_points]
        input_neighbors += [conv_i]
        input_pools += [pool_i]
        input_upsamples += [up_i]
        input_batches_len += [batched_lengths]
        batched_points, batched_lengths = pool_p, pool_b
        r_normal *= 2
        layer += 1
        layer_blocks = []
    if deferred:
        widths = point_ops.finish_radius_tables_async(deferred, flags_all)
        yield widths
        for (lst, i), t in zip(slots, widths.finish()):
            lst[i] = t
    out = {'points': input_points, 'neighbors': input_neighbors, 'pools': input_pools, 'upsamples': input_upsamples,
           'features': batched_features, 'stack_lengths': input_batches_len}
    if len(list_data) > 1:
        out['pair_rows'] = {}
        for lens in input_batches_len:
            ends = np.cumsum(lens.numpy().astype(np.int64))
            out['pair_rows'][int(ends[-1])] = [0] + [int(v) for v in ends[1::2]]
        return out
    for key, val in zip(('rot', 'trans', 'correspondences', 'src_pcd_raw', 'tgt_pcd_raw', 'src_nghb', 'tgt_nghb',
                         'sample'), item[4:]):
        out[key] = val
    return out


def calibrate_neighbors(dataset, config, collate_fn=collate_fn_descriptor, keep_ratio=0.8, samples_threshold=2000):
    """80-th percentile of the neighbour-count histogram per layer (dataloader.py:200-232)."""
    hist_n = int(np.ceil(4 / 3 * np.pi * (config.deform_radius + 1) ** 3))
    neighb_hists = np.zeros((config.num_layers, hist_n), dtype=np.int32)
    for i in range(len(dataset)):
        batched_input = collate_fn([dataset[i]], config, neighborhood_limits=[hist_n] * 5)
        counts = [torch.sum(m < m.shape[0], dim=1).cpu().numpy() for m in batched_input['neighbors']]
        hists = [np.bincount(c, minlength=hist_n)[:hist_n] for c in counts]
        neighb_hists += np.vstack(hists)
        if np.min(np.sum(neighb_hists, axis=1)) > samples_threshold:
            break
    cumsum = np.cumsum(neighb_hists.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0)
