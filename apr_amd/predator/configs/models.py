"""Block lists of the KPFCNN variants (Predator_APR/configs/models.py:1-77)."""
_ENC = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb',
        'resnetb_strided', 'resnetb', 'resnetb']
_DEC = ['nearest_upsample', 'unary', 'nearest_upsample', 'unary', 'nearest_upsample', 'last_unary']
architectures = dict()
architectures['indoor'] = _ENC + _DEC
architectures['kitti'] = _ENC + _DEC
architectures['nuscenes'] = _ENC + _DEC
architectures['modelnet'] = ['simple', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided',
                             'resnetb', 'resnetb', 'nearest_upsample', 'unary', 'unary', 'nearest_upsample', 'unary',
                             'last_unary']


class Config(dict):
    """Attribute dict (the reference uses easydict, absent here)."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def kitti_config(**over):
    """Model + overlap-attention sections of Predator_APR/configs/test/kitti.yaml:12-39."""
    cfg = Config(num_layers=4, in_points_dim=3, first_feats_dim=256, final_feats_dim=32, first_subsampling_dl=0.3,
                 in_feats_dim=1, conv_radius=4.25, deform_radius=5.0, num_kernel_points=15, KP_extent=2.0,
                 KP_influence='linear', aggregation_mode='sum', fixed_kernel_points='center', use_batch_norm=True,
                 batch_norm_momentum=0.02, deformable=False, modulated=False, add_cross_score=True,
                 condition_feature=True, gnn_feats_dim=256, dgcnn_k=10, num_head=4, nets=['self', 'cross', 'self'],
                 architecture=architectures['kitti'], switch_to_decoder=False, symmetric=False,
                 point_generation_ratio=4)
    cfg.update(over)
    return cfg
