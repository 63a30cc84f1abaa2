"""Dump the MultiTalent region tables (constant DATA: region name -> label tuple, output index, valid regions per
source dataset; Task100_MultiTalent.py:35-207) from the imported reference into a JSON data file shipped with the
package.  Runs only in the build container."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import
ref_import.install()
from nnunet.dataset_conversion import Task100_MultiTalent as T

out = {
    'MultiTalent_regions': {k: list(v) for k, v in T.MultiTalent_regions.items()},
    'MultiTalent_region_output_idx_mapping': dict(T.MultiTalent_region_output_idx_mapping),
    'MultiTalent_valid_regions': {k: list(v) for k, v in T.MultiTalent_valid_regions.items()},
    'MultiTalent_regions_class_order': {k: list(v) for k, v in T.MultiTalent_regions_class_order.items()},
    'MultiTalent_task_ids': list(T.MultiTalent_task_ids) if not isinstance(T.MultiTalent_task_ids, dict) else T.MultiTalent_task_ids,
    'MultiTalent_labels': {str(k): v for k, v in T.MultiTalent_labels.items()} if isinstance(T.MultiTalent_labels, dict) else list(T.MultiTalent_labels),
    'MultiTalent_task_label_maps': {str(k): ({str(a): b for a, b in v.items()} if isinstance(v, dict) else v)
                                    for k, v in T.MultiTalent_task_label_maps.items()},
}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'multitalent_amd', 'dataset_conversion', 'multitalent_tables.json')
json.dump(out, open(dst, 'w'), indent=1)
print('wrote', os.path.normpath(dst))
