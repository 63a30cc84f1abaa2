"""Dump the hot-path-relevant entries of the reference's two north-star plans files (stage 1 = 3d_fullres, plus the CT intensity
statistics and the top-level keys make_plans restates) to tests/golden/plans_stage1.json: data read from the reference's pickles, so
that a CPU test can pin multitalent_amd/plans.py (which restates them as constants) without the reference checkout.
Run in the BUILD container:  python tools/oracle_gen/dump_plans_stage1.py"""
import json
import os
import pickle

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/MultiTalent_plans'


def plain(v):
    if isinstance(v, np.ndarray):
        return [plain(i) for i in v.tolist()]
    if isinstance(v, dict):
        return {str(k): plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [plain(i) for i in v]
    if isinstance(v, (np.bool_,)):
        return bool(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    return repr(v)


def main():
    out = {}
    for tag, fname in (('plain', 'MultiTalent_bs4_plans_3D.pkl'), ('resenc', 'MultiTalent_resenc_bs4_plans_3D.pkl')):
        with open(os.path.join(REF, fname), 'rb') as f:
            p = pickle.load(f)
        st = p['plans_per_stage'][1]
        e = {'stage': {k: plain(st[k]) for k in sorted(st) if k in ('batch_size', 'patch_size', 'num_pool_per_axis', 'pool_op_kernel_sizes', 'conv_kernel_sizes',
                                                                      'do_dummy_2D_data_aug', 'current_spacing', 'num_blocks_encoder', 'num_blocks_decoder',
                                                                      'original_spacing', 'median_patient_size_in_voxels')},
             'num_stages': len(p['plans_per_stage']),
             'top': {k: plain(p[k]) for k in ('num_modalities', 'modalities', 'normalization_schemes', 'num_classes', 'base_num_features', 'use_mask_for_norm',
                                              'transpose_forward', 'transpose_backward', 'data_identifier', 'conv_per_stage', 'preprocessor_name') if k in p},
             'ct_stats': {k: plain(v) for k, v in p['dataset_properties']['intensityproperties'][0].items() if k in ('mean', 'sd', 'percentile_00_5', 'percentile_99_5')}}
        out[tag] = e
    dst = os.path.join(ROOT, 'tests', 'golden', 'plans_stage1.json')
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
