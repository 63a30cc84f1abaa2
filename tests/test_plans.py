"""multitalent_amd/plans.py restates stage 1 (3d_fullres) of the reference's two plans files (MultiTalent_plans/*.pkl) as constants;
tests/golden/plans_stage1.json holds the same entries as READ FROM THE REFERENCE'S PICKLES (tools/oracle_gen/dump_plans_stage1.py)."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _eq(a, b):
    if isinstance(a, np.ndarray):
        a = a.tolist()
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_eq(x, y) for x, y in zip(a, b))
    if isinstance(a, float) or isinstance(b, float):
        return abs(float(a) - float(b)) < 1e-12
    return a == b


def test_restated_plans_equal_the_reference_pickles():
    from multitalent_amd import plans as P
    ref = json.load(open(os.path.join(G, 'plans_stage1.json')))
    for tag, const in (('plain', P.TASK100_PLAIN_STAGE), ('resenc', P.TASK100_RESENC_STAGE)):
        st = ref[tag]['stage']
        for k, v in const.items():
            assert k in st, (tag, k)
            assert _eq(v, st[k]), (tag, k, v, st[k])
        # nothing the hot path reads is missing from the restatement
        for k in ('batch_size', 'patch_size', 'pool_op_kernel_sizes', 'conv_kernel_sizes', 'do_dummy_2D_data_aug', 'current_spacing'):
            assert k in const, (tag, k)
        assert ref[tag]['num_stages'] == 2                       # stage 1 IS the last stage (3d_fullres)
    for k, v in P.TASK100_CT_STATS.items():
        assert abs(v - ref['plain']['ct_stats'][k]) < 5e-3 * max(1.0, abs(v)), (k, v, ref['plain']['ct_stats'][k])
    # the dict make_plans builds carries the reference's top-level entries
    mp = P.make_plans(P.TASK100_PLAIN_STAGE)
    top = ref['plain']['top']
    for k in ('num_modalities', 'num_classes', 'base_num_features', 'transpose_forward', 'transpose_backward', 'conv_per_stage', 'preprocessor_name'):
        assert _eq(mp[k], top[k]), (k, mp[k], top[k])
    assert mp['data_identifier'] == top['data_identifier']
    assert {int(k): v for k, v in top['normalization_schemes'].items()} == mp['normalization_schemes']
    assert {int(k): v for k, v in top['use_mask_for_norm'].items()} == mp['use_mask_for_norm']
