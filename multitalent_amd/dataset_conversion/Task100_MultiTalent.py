"""MultiTalent region tables (loss semantics of the hot path).  Same names as the reference module
nnunet/dataset_conversion/Task100_MultiTalent.py:118-207; the values are constant data stored in
multitalent_tables.json (dumped by tools/oracle_gen/dump_region_tables.py).  The dataset conversion code
of the reference (offline data prep) is out of scope."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'multitalent_tables.json')) as _f:
    _T = json.load(_f)

# region name -> tuple of label values (47 regions; 3 are unions: 03_liver, 07_pancreas, 64_both_kidneys)
MultiTalent_regions = {k: tuple(v) for k, v in _T['MultiTalent_regions'].items()}
# region name -> network output channel (= enumeration order)
MultiTalent_region_output_idx_mapping = dict(_T['MultiTalent_region_output_idx_mapping'])
# source dataset -> regions that are annotated in it
MultiTalent_valid_regions = {k: tuple(v) for k, v in _T['MultiTalent_valid_regions'].items()}
MultiTalent_regions_class_order = {k: tuple(v) for k, v in _T['MultiTalent_regions_class_order'].items()}
MultiTalent_task_ids = _T['MultiTalent_task_ids']
MultiTalent_labels = _T['MultiTalent_labels']
MultiTalent_task_label_maps = _T['MultiTalent_task_label_maps']


def region_label_lut():
    """64-bit mask over label values for each output channel (consumed by mt_multitalent_loss_*)."""
    lut = [0] * len(MultiTalent_regions)
    for name, labels in MultiTalent_regions.items():
        m = 0
        for l in labels:
            assert 0 <= l < 64
            m |= (1 << l)
        lut[MultiTalent_region_output_idx_mapping[name]] = m
    return lut


def valid_mask(region_names):
    m = 0
    for r in region_names:
        m |= (1 << MultiTalent_region_output_idx_mapping[r])
    return m
