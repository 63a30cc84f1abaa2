"""Plans handling (boundary seam 3): the reference's plans `.pkl` is a pickled dict written by its
experiment planner (experiment_planner_baseline_3DUNet.py:341-354) and read at nnUNetTrainer.py:319-392.  The two
north-star plans' hot-path-relevant entries are restated here as constants (the shipped pkl embeds cluster paths and
390 KB of per-case statistics); `load_plans_file` reads any plans pkl with plain pickle."""
import pickle

import numpy as np

# MultiTalent_plans/MultiTalent_bs4_plans_3D.pkl, plans_per_stage[1] (3d_fullres = last stage)
TASK100_PLAIN_STAGE = {
    'batch_size': 4, 'patch_size': np.array([96, 192, 192]), 'num_pool_per_axis': [4, 5, 5],
    'pool_op_kernel_sizes': [[2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]],
    'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False, 'current_spacing': np.array([1.5, 1.0, 1.0]),
}
# MultiTalent_plans/MultiTalent_resenc_bs4_plans_3D.pkl, plans_per_stage[1]
TASK100_RESENC_STAGE = {
    'batch_size': 2, 'patch_size': np.array([96, 192, 192]),
    'pool_op_kernel_sizes': [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
    'conv_kernel_sizes': [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
    'num_blocks_encoder': [1, 2, 3, 4, 4, 4], 'num_blocks_decoder': [1, 1, 1, 1, 1],
    'do_dummy_2D_data_aug': False, 'current_spacing': np.array([1.5, 1.0, 1.0]),
}
# global CT intensity statistics of the Task100 plan (dataset_properties.intensityproperties[0]); used for the
# clip + z-score normalisation of CT (preprocessing.py:275-285) and for the synthetic benchmark inputs
TASK100_CT_STATS = {'mean': 63.44, 'sd': 175.48, 'percentile_00_5': -927.0, 'percentile_99_5': 275.0}


def make_plans(stage_plan, base_num_features=30, num_modalities=1, num_classes=47, conv_per_stage=2, stage=1):
    return {
        'num_stages': stage + 1, 'num_modalities': num_modalities, 'modalities': {0: 'CT'},
        'normalization_schemes': {0: 'CT'}, 'num_classes': num_classes, 'all_classes': list(range(1, num_classes + 1)),
        'base_num_features': base_num_features, 'use_mask_for_norm': {0: False}, 'transpose_forward': [0, 1, 2],
        'transpose_backward': [0, 1, 2], 'data_identifier': 'MultiTalent_data', 'conv_per_stage': conv_per_stage,
        'plans_per_stage': {stage: dict(stage_plan)}, 'preprocessor_name': 'GenericPreprocessor',
        'dataset_properties': {'intensityproperties': {0: dict(TASK100_CT_STATS)}},
    }


def load_plans_file(fname):
    with open(fname, 'rb') as f:
        return pickle.load(f)
