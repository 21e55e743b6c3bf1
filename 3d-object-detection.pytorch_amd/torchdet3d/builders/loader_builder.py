"""Loader boundary.  The reference's Objectron dataset + albumentations pipeline
(torchdet3d/builders/loader_builder.py:14-36, dataloaders/objectron_main.py) is CPU data preparation and out of
scope for this build; what the hot path needs is its OUTPUT CONTRACT
(objectron_main.py:51-96 + utils/transforms.py:103-114): batches `(imgs f32 [B,3,H,W] normalised, gt_kp f32
[B,9,2] in [0,1], gt_cats int64 [B])`.  `SyntheticCrops` produces exactly that; `build_loader` serves it when
`cfg.data.root == 'synthetic'`.

One process per GPU (an unchanged scripts/main.py under `python -m torch.distributed.run`; `build_model`, called first by
main.py:46, has joined the process group by the time main.py:63 builds the loaders): the reference's `nn.DataParallel`
scatters ONE batch of `train_batch_size` crops over the replicas (main.py:60-61), so here every rank draws its own
`train_batch_size / world` share of the same global batch -- `DistributedSampler` over the one dataset every rank holds
(same shuffle seed, disjoint index sets, `set_epoch` called by `Trainer.train`) -- and validation walks the samples
`rank, rank + world, ...` without padding (the per-rank partial sums are all-reduced by `Evaluator.val`)."""
import torch

from ..parallel import rank, world_size


class SyntheticCrops(torch.utils.data.Dataset):
    def __init__(self, n, size=(224, 224), num_classes=9, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.imgs = torch.randn(n, 3, size[1], size[0], generator=g)
        self.kp = torch.rand(n, 9, 2, generator=g)
        self.cats = torch.randint(0, max(num_classes, 1), (n,), generator=g)

    def __len__(self):
        return self.imgs.shape[0]

    def __getitem__(self, i):
        return self.imgs[i], self.kp[i], self.cats[i]


def build_loader(config, mode='train'):
    if config.data.root != 'synthetic':
        raise NotImplementedError('only the synthetic crop source is built (cfg.data.root = "synthetic"); the '
                                  'Objectron JSON/albumentations loader is out of scope')
    n = config.data.synthetic_len or 64
    size = tuple(config.data.resize) if config.data.resize else (224, 224)
    mk = lambda seed: SyntheticCrops(n, size, config.model.num_classes or 9, seed)
    tb, vb = config.data.train_batch_size or 8, config.data.val_batch_size or 8
    world, rk = world_size(), rank()
    if world == 1:
        train = torch.utils.data.DataLoader(mk(1), batch_size=tb, shuffle=True, drop_last=True)
        val = torch.utils.data.DataLoader(mk(2), batch_size=vb, shuffle=False)
    else:
        if tb % world:
            raise ValueError(f'data.train_batch_size = {tb} is the GLOBAL batch (scripts/main.py:60-61 scatters it over the '
                             f'replicas): it must be divisible by the {world} ranks of this launch')
        ds = mk(1)
        seed = int(getattr(getattr(config, 'utils', None), 'random_seeds', 0) or 0)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rk, shuffle=True, seed=seed,
                                                                  drop_last=True)
        train = torch.utils.data.DataLoader(ds, batch_size=tb // world, sampler=sampler, drop_last=True)
        dv = mk(2)
        val = torch.utils.data.DataLoader(torch.utils.data.Subset(dv, range(rk, len(dv), world)),
                                          batch_size=max(vb // world, 1), shuffle=False)
    test = torch.utils.data.DataLoader(mk(3), batch_size=1, shuffle=False)
    return train, val, test
