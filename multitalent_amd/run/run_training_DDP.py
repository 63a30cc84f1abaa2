"""`run_training_DDP` (reference run/run_training_DDP.py:30-200): one process per GPU, launched by `torch.distributed.run`
(LOCAL_RANK from the environment or --local_rank).  Flow: configuration -> trainer(plans, fold, local_rank, ...) ->
initialize -> [continue | pretrained weights] -> run_training -> validate.  Flags that select subsystems outside the hot path
(`--find_lr`, cascade next-stage prediction, postprocessing search) are accepted and ignored."""
import argparse
import os

from ..training.network_training.nnUNetTrainer import nnUNetTrainer
from .default_configuration import convert_id_to_task_name, default_plans_identifier, get_default_configuration
from .load_pretrained_weights import load_pretrained_weights


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("network")
    ap.add_argument("network_trainer")
    ap.add_argument("task", help="can be task name or task id")
    ap.add_argument("fold", help="0, 1, ..., 5 or 'all'")
    ap.add_argument("-val", "--validation_only", action="store_true")
    ap.add_argument("-c", "--continue_training", action="store_true")
    ap.add_argument("-p", default=default_plans_identifier)
    ap.add_argument("--use_compressed_data", default=False, action="store_true")
    ap.add_argument("--deterministic", default=False, action="store_true")
    ap.add_argument("--local_rank", default=None, type=int)
    ap.add_argument("--local-rank", dest='local_rank', type=int)
    ap.add_argument("--fp32", default=False, action="store_true")
    ap.add_argument("--dbs", default=False, action="store_true")
    ap.add_argument("--npz", default=False, action="store_true")
    ap.add_argument("--valbest", default=False, action="store_true")
    ap.add_argument("--find_lr", default=False, action="store_true")
    ap.add_argument("--val_folder", default="validation_raw")
    ap.add_argument("--disable_saving", action='store_true')
    ap.add_argument('--val_disable_overwrite', action='store_false', default=True)
    ap.add_argument('--disable_next_stage_pred', action='store_true', default=False)
    ap.add_argument("--disable_postprocessing_on_folds", action='store_true')
    ap.add_argument('-pretrained_weights', type=str, default=None)
    a = ap.parse_args(argv)
    task = a.task if a.task.startswith("Task") else convert_id_to_task_name(int(a.task))
    fold = a.fold if a.fold == 'all' else int(a.fold)
    local_rank = a.local_rank if a.local_rank is not None else int(os.environ.get('LOCAL_RANK', 0))
    plans_file, output_folder_name, dataset_directory, batch_dice, stage, trainer_class = \
        get_default_configuration(a.network, task, a.network_trainer, a.p)
    if trainer_class is None:
        raise RuntimeError("Could not find trainer class in multitalent_amd.training.network_training")
    assert issubclass(trainer_class, nnUNetTrainer), "network_trainer was found but is not derived from nnUNetTrainer"
    trainer = trainer_class(plans_file, fold, local_rank, output_folder=output_folder_name, dataset_directory=dataset_directory,
                            batch_dice=batch_dice, stage=stage, unpack_data=not a.use_compressed_data,
                            deterministic=a.deterministic, fp16=not a.fp32, distribute_batch_size=a.dbs)
    if a.disable_saving:
        trainer.save_latest_only = False
        trainer.save_intermediate_checkpoints = False
        trainer.save_best_checkpoint = False
        trainer.save_final_checkpoint = False
    trainer.initialize(not a.validation_only)
    if not a.validation_only:
        if a.continue_training:
            trainer.load_latest_checkpoint()
        elif a.pretrained_weights is not None:
            load_pretrained_weights(trainer.network, a.pretrained_weights)
        trainer.run_training()
    elif a.valbest:
        trainer.load_best_checkpoint(train=False)
    else:
        trainer.load_final_checkpoint(train=False)
    trainer.network.eval()
    trainer.validate(save_softmax=a.npz, validation_folder_name=a.val_folder,
                     run_postprocessing_on_folds=not a.disable_postprocessing_on_folds, overwrite=a.val_disable_overwrite)


if __name__ == "__main__":
    main()
