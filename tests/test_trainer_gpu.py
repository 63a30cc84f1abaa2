"""Trainer plugin surface on the GPU: discovery by name, plans parsing, run_iteration on the fused hot loop,
checkpoint round trip in the reference's file format, restore_model, sliding-window prediction wrapper."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def small_plans(resenc=False):
    from multitalent_amd import plans as P
    if resenc:
        sp = {'batch_size': 2, 'patch_size': np.array([16, 32, 32]),
              'pool_op_kernel_sizes': [[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 2]],
              'conv_kernel_sizes': [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
              'num_blocks_encoder': [1, 2, 2, 2], 'num_blocks_decoder': [1, 1, 1], 'do_dummy_2D_data_aug': False}
    else:
        sp = {'batch_size': 2, 'patch_size': np.array([16, 32, 32]), 'pool_op_kernel_sizes': [[2, 2, 2], [2, 2, 2], [1, 2, 2]],
              'conv_kernel_sizes': [[3, 3, 3]] * 4, 'do_dummy_2D_data_aug': False}
    return P.make_plans(sp, base_num_features=8, num_classes=47, stage=1)


@pytest.fixture(scope='module')
def pg():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='env://')
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("name,resenc", [("MultiTalent_trainer_ddp", False), ("nnUNetTrainerV2_MultiTalent", False),
                                         ("MultiTalent_trainer_resenc_ddp", True)])
def test_multitalent_trainer_roundtrip(dev, pg, tmp_path, name, resenc):
    from multitalent_amd.training.model_restore import find_trainer_class, restore_model
    cls = find_trainer_class(name)
    tr = cls(small_plans(resenc), 0, 0, output_folder=str(tmp_path), stage=1)
    tr.initialize(True)
    assert tr.num_classes == 47 and tr.regions_class_order == list(range(47))
    gen = tr._default_generator()
    tr.network.train()
    losses = [tr.run_iteration(gen, True) for _ in range(3)]
    assert all(np.isfinite(l[0]) for l in losses) and len(losses[0]) == 3
    assert losses[2][0] < losses[0][0]          # a fixed batch must be learnable
    with torch.no_grad():
        tr.network.eval()
        v = tr.run_iteration(gen, False, True)
    assert np.isfinite(v[0]) and len(tr.online_eval_tp) == 1
    # L4 (MultiTalent_Trainer_DDP.py:372-410): [B][47] hard tp/fp/fn of the full-resolution output == the oracle's restatement on
    # the same logits (exact counts except for voxels whose logit is within 1e-5 of the decision boundary)
    from oracle import reference_ops as R
    from multitalent_amd.dataset_conversion.Task100_MultiTalent import MultiTalent_region_output_idx_mapping, MultiTalent_regions
    d = next(gen)
    with torch.no_grad():
        logits = tr.network(tr._to_device(d['data']))[0].float().cpu()
    tgt, valid = tr.loss_args(d)
    otp, ofp, ofn, _ = R.multitalent_online_evaluation(logits, tgt[0].cpu(), valid, MultiTalent_regions, MultiTalent_region_output_idx_mapping)
    near = (logits.abs() < 1e-5).flatten(2).sum(2).numpy()                         # [B, 47] undecidable voxels
    B = logits.shape[0]
    assert len(tr.online_eval_tp[0]) == B and len(tr.online_eval_tp[0][0]) == 47
    for got, ref in ((tr.online_eval_tp[0], otp), (tr.online_eval_fp[0], ofp), (tr.online_eval_fn[0], ofn)):
        assert (np.abs(np.array(got) - ref) <= near).all()
    assert np.array(tr.online_eval_tp[0]).sum() + np.array(tr.online_eval_fp[0]).sum() + np.array(tr.online_eval_fn[0]).sum() > 0
    tr.finish_online_evaluation()
    assert len(tr.all_val_eval_metrics) == 1 and 0.0 <= tr.all_val_eval_metrics[0] <= 1.0 and tr.online_eval_tp == []
    f = os.path.join(str(tmp_path), 'model_latest.model')
    tr.save_checkpoint(f)
    ck = torch.load(f, map_location='cpu', weights_only=False)
    assert set(ck) >= {'epoch', 'state_dict', 'optimizer_state_dict', 'lr_scheduler_state_dict', 'plot_stuff', 'best_stuff'}
    info = pickle.load(open(f + '.pkl', 'rb'))
    assert info['name'] == name and set(info) == {'init', 'name', 'class', 'plans'}
    assert len(ck['optimizer_state_dict']['state']) == len(list(tr.network.parameters()))
    # DDP checkpoints carry a 'module.' prefix: must be stripped on load (nnUNetTrainerV2_DDP.py:650-662)
    ck['state_dict'] = {'module.' + k: v for k, v in ck['state_dict'].items()}
    tr2 = restore_model(f + '.pkl')
    tr2.initialize(False)
    tr2.load_checkpoint_ram(ck, False)
    vol = np.random.RandomState(0).randn(1, 20, 40, 40).astype(np.float32)
    s1, p1 = tr.predict_preprocessed_data_return_seg_and_softmax(vol, do_mirroring=True, mirror_axes=(0, 1, 2), verbose=False)
    s2, p2 = tr2.predict_preprocessed_data_return_seg_and_softmax(vol, do_mirroring=True, mirror_axes=(0, 1, 2), verbose=False)
    assert p1.shape == (47, 20, 40, 40) and s1.shape == (20, 40, 40)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    assert tr.network.training is False         # the wrapper restores the mode it found (eval, set above)


def test_single_gpu_trainer_softmax(dev, tmp_path):
    from multitalent_amd import plans as P
    from multitalent_amd.training.model_restore import find_trainer_class
    sp = {'batch_size': 2, 'patch_size': np.array([16, 32, 32]), 'pool_op_kernel_sizes': [[2, 2, 2], [2, 2, 2], [1, 2, 2]],
          'conv_kernel_sizes': [[3, 3, 3]] * 4, 'do_dummy_2D_data_aug': False}
    plans = P.make_plans(sp, base_num_features=8, num_classes=1, stage=0)
    tr = find_trainer_class('nnUNetTrainerV2')(plans, 0, output_folder=str(tmp_path), batch_dice=False, stage=0)
    tr.initialize(True)
    assert tr.num_classes == 2
    gen = tr._default_generator()
    tr.network.train()
    l = [float(tr.run_iteration(gen, True)) for _ in range(4)]
    assert l[-1] < l[0]
    tr.maybe_update_lr(500)
    assert abs(tr.train_step.lr - 1e-2 * (1 - 500 / 1000) ** 0.9) < 1e-12


def test_device_batch_feeder_matches_cpu_pipeline(dev):
    """DeviceBatchFeeder: pinned double-buffered upload + on-device label pyramid == the reference's CPU result
    (oracle restatement of DownsampleSegForDSTransform2 + RemoveLabelTransform), order preserved, pyramids passed through."""
    import numpy as np
    from oracle import reference_ops as R
    from multitalent_amd.training.dataloading.device_feed import DeviceBatchFeeder
    rng = np.random.RandomState(9)
    scales = [[1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.25, 0.25]]
    shape = (16, 32, 48)
    batches = [{'data': rng.randn(2, 1, *shape).astype(np.float32),
                'target': rng.randint(-1, 48, size=(2, 1) + shape).astype(np.float32),
                'properties': [{'valid_regions': ['03_liver']}] * 2, 'keys': ['a', 'b']} for _ in range(5)]
    got = list(DeviceBatchFeeder(iter(batches), ds_scales=scales))
    assert len(got) == 5
    for b, g in zip(batches, got):
        assert g['data'].is_cuda and np.array_equal(g['data'].cpu().numpy(), b['data'])
        ref = R.downsample_seg_for_ds_transform2(R.remove_label(b['target']), scales, 0)
        assert len(g['target']) == 3
        for t, r in zip(g['target'], ref):
            assert np.array_equal(t.cpu().numpy(), r)
        assert g['properties'] is b['properties'] and g['keys'] == b['keys']
    # an already-built pyramid is uploaded level by level, untouched
    pyr = [{'data': batches[0]['data'], 'target': [np.ones((2, 1, 4, 4, 4), np.float32), np.zeros((2, 1, 2, 2, 2), np.float32)]}]
    g = next(DeviceBatchFeeder(iter(pyr)))
    assert [tuple(t.shape) for t in g['target']] == [(2, 1, 4, 4, 4), (2, 1, 2, 2, 2)] and float(g['target'][0].sum()) == 128.0


def test_training_from_preprocessed_cases_on_disk(dev, pg, tmp_path):
    """SURVEY §8f rank 2 end to end: <dataset_directory>/<data_identifier>_stage1/*.npz|pkl -> unpack -> split -> sqrt-balanced
    DataLoader3D -> one label map per batch -> device-side pyramid -> fused MultiTalent loss/step, through run_training()."""
    from multitalent_amd import plans as P
    from multitalent_amd.training.model_restore import find_trainer_class
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loader.npz'))
    sp = {'batch_size': 2, 'patch_size': np.array([16, 32, 32]), 'pool_op_kernel_sizes': [[2, 2, 2], [2, 2, 2]],
          'conv_kernel_sizes': [[3, 3, 3]] * 3, 'do_dummy_2D_data_aug': False}
    plans = P.make_plans(sp, base_num_features=8, num_classes=47, stage=1)
    folder = tmp_path / (plans['data_identifier'] + '_stage1')
    folder.mkdir()
    for k in z.files:
        if k.startswith('case/'):
            arr = z[k]
            np.savez_compressed(str(folder / (k[5:] + '.npz')), data=arr)
            props = {'class_locations': {c: np.argwhere(arr[-1] == c) for c in (1, 2, 4)},
                     'valid_regions': ('03_liver', '03_cancer') if k[5:].startswith('BTCV') else ('07_pancreas',)}
            with open(str(folder / (k[5:] + '.pkl')), 'wb') as f:
                pickle.dump(props, f)
    tr = find_trainer_class('MultiTalent_trainer_ddp')(plans, 'all', 0, output_folder=str(tmp_path / 'out'),
                                                        dataset_directory=str(tmp_path), stage=1)
    tr.initialize(True)
    tr.max_num_epochs, tr.num_batches_per_epoch, tr.num_val_batches_per_epoch, tr.save_every = 2, 3, 1, 100
    np.random.seed(0)
    tr.run_training()
    assert type(tr.tr_gen).__name__ == 'MoreDADeviceAugmenter' and type(tr.val_gen).__name__ == 'SegToTargetGenerator'
    assert tuple(tr.basic_generator_patch_size) == (35, 51, 42)          # loader patch of the 16x32x32 network patch (+-30 deg, /0.85)
    b = next(tr.tr_gen)
    assert b['data'].is_cuda and tuple(b['data'].shape) == (2, 1, 16, 32, 32) and tuple(b['target'].shape) == (2, 1, 16, 32, 32)
    assert any(f.endswith('.npy') for f in os.listdir(str(folder)))                    # unpacked for memory-mapped reads
    assert len(tr.all_tr_losses) == 2 and np.isfinite(tr.all_tr_losses).all() and np.isfinite(tr.all_val_losses).all()
    assert os.path.isfile(str(tmp_path / 'out' / 'all' / 'model_final_checkpoint.model'))      # <output_folder>/all (nnUNetTrainer.py:134-152)
