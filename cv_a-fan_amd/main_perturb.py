"""A-FAN training entry point — same flags, stdout lines and output files as the reference's
Classification/main_perturb.py (flags :28-49, loop :97-150, train :153-225, validate :227-263), so
`bash cmd/run_perturb.sh` keeps working.  Additions (all optional): --arch, --dtype, --synthetic,
--max_iters; launched under torch.distributed.run it trains data parallel, one rank per MI355X.

What differs from the reference is execution only: the iteration body is train_step.AfanTrainer.step
(HIP kernels, no host sync), metrics stay on the device and are read back every --print_freq iterations,
the perturbation norms come out of the last PGD kernel instead of a host-side reduction.
"""
import argparse
import os
import pickle
import random
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

if __package__ in (None, ""):  # executed as a script (cmd/run_perturb.sh): import the hyphenated package by path
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module("cv_a-fan_amd")
    resnet_s, train_step, host = _pkg.resnet_s, _pkg.train_step, _pkg.host
else:
    from . import host, resnet_s, train_step

parser = argparse.ArgumentParser(description="A-FAN CIFAR-10 training on MI355X")
# ---- base setting (main_perturb.py:28-33)
parser.add_argument("--data", type=str, default="../data", help="location of the data corpus (cifar-10-batches-py)")
parser.add_argument("--print_freq", default=50, type=int, help="print frequency")
parser.add_argument("--seed", default=None, type=int, help="random seed")
parser.add_argument("--gpu", type=int, default=0, help="gpu device id")
parser.add_argument("--resume", action="store_true", help="resume from checkpoint")
parser.add_argument("--save_dir", help="The directory used to save the trained models", default="res56s_adv_aug", type=str)
# ---- optimizer setting (main_perturb.py:36-41)
parser.add_argument("--batch_size", type=int, default=128, help="batch size (global; split across ranks)")
parser.add_argument("--lr", default=0.1, type=float, help="initial learning rate")
parser.add_argument("--momentum", default=0.9, type=float, help="momentum")
parser.add_argument("--weight_decay", default=5e-4, type=float, help="weight decay")
parser.add_argument("--epochs", default=200, type=int, help="number of total epochs to run")
parser.add_argument("--decreasing_lr", default="50,150", help="decreasing strategy")
# ---- A-FAN setting (main_perturb.py:44-49)
parser.add_argument("--steps", default=5, type=int, help="PGD-steps")
parser.add_argument("--perturb_idx", help="index of perturb layers", default=13, type=int)
parser.add_argument("--gamma", help="index of PGD gamma", default=1.5, type=float)
parser.add_argument("--eps", default=2, type=float)
parser.add_argument("--randinit", action="store_true", help="whether using randinit")
parser.add_argument("--clip", action="store_true", help="whether using clip")
# ---- additions
parser.add_argument("--arch", default="resnet56s", choices=sorted(resnet_s.ARCHS))
parser.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="backbone compute dtype")
parser.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"],
                    help="internal activation / weight layout (nhwc: the library's MFMA convolutions; nchw: the general fp32-arithmetic kernels)")
parser.add_argument("--dual_bn", action="store_true", help="auxiliary BatchNorm set for adversarial features (not in the "
                    "reference: extra state_dict keys <bn>.adv.*; evaluation uses the main set)")
parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images instead of CIFAR-10")
parser.add_argument("--max_iters", type=int, default=0, help="stop each epoch after this many iterations (0 = all)")


def setup_seed(seed):
    """main_perturb.py:310-315 (cudnn.deterministic selects deterministic MIOpen algorithms on ROCm)."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.backends.cudnn.deterministic = True


class AverageMeter(object):
    """main_perturb.py:271-286"""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


# ------------------------------------------------------------------------------------------------ data
def _load_cifar10(root):
    """cifar-10-batches-py pickles -> uint8 NCHW arrays; the 45k/5k train/val split of dataset.py:43-45."""
    d = root if os.path.basename(root.rstrip("/")) == "cifar-10-batches-py" else os.path.join(root, "cifar-10-batches-py")
    xs, ys = [], []
    for i in range(1, 6):
        with open(os.path.join(d, f"data_batch_{i}"), "rb") as f:
            b = pickle.load(f, encoding="latin1")
        xs.append(np.asarray(b["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(b["labels"], dtype=np.int64))
    with open(os.path.join(d, "test_batch"), "rb") as f:
        b = pickle.load(f, encoding="latin1")
    xt, yt = np.asarray(b["data"], dtype=np.uint8).reshape(-1, 3, 32, 32), np.asarray(b["labels"], dtype=np.int64)
    x, y = np.concatenate(xs), np.concatenate(ys)
    return (x[:45000], y[:45000]), (x[45000:], y[45000:]), (xt, yt)


class DeviceLoader:
    """Whole split resident in HBM as uint8 (CIFAR-10 train = 138 MB of 288 GB); per batch: shuffle index, random
    crop (pad 4) + horizontal flip (dataset.py:36-39) and the /255 ToTensor scaling run on the device."""

    def __init__(self, x_u8, y, batch, device, train, rank=0, world=1, drop_last=True, seed=None):
        self.x = torch.as_tensor(x_u8).to(device)
        self.y = torch.as_tensor(y).to(device)
        self.batch, self.train, self.rank, self.world, self.device = batch, train, rank, world, device
        # data parallel: every rank must slice the SAME permutation (its own CPU generator would give overlapping shards):
        # a generator seeded with (seed + epoch), `seed` agreed on by all ranks (main() broadcasts rank 0's draw)
        self.seed, self.epoch = seed, 0
        n = self.x.shape[0]
        self.n_batches = n // batch if drop_last else (n + batch - 1) // batch

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        n = self.x.shape[0]
        if not self.train:
            perm = torch.arange(n)
        elif self.world > 1:
            if self.seed is None:
                raise RuntimeError("a data-parallel DeviceLoader needs a seed shared by all ranks")
            perm = torch.randperm(n, generator=torch.Generator().manual_seed(int(self.seed) + self.epoch))
            self.epoch += 1
        else:
            perm = torch.randperm(n)                                  # CPU generator, like DataLoader's sampler
        per = self.batch // self.world
        for b in range(self.n_batches):
            idx = perm[b * self.batch:(b + 1) * self.batch]
            idx = idx[self.rank * per:(self.rank + 1) * per] if self.world > 1 else idx
            idx = idx.to(self.device)
            x, y = self.x[idx], self.y[idx]
            if self.train:
                m = x.shape[0]
                xp = torch.nn.functional.pad(x, (4, 4, 4, 4))                         # [m, 3, 40, 40]
                ar = torch.arange(32, device=self.device)
                rows = torch.randint(0, 9, (m,), device=self.device)[:, None] + ar[None, :]
                cols = torch.randint(0, 9, (m,), device=self.device)[:, None] + ar[None, :]
                flip = torch.rand(m, device=self.device) < 0.5
                cols = torch.where(flip[:, None], cols.flip(1), cols)
                bi = torch.arange(m, device=self.device)[:, None, None]
                x = xp[bi, :, rows[:, :, None], cols[:, None, :]].permute(0, 3, 1, 2).contiguous()
            yield x.float().div_(255.0), y


class SyntheticLoader:
    """U[0,1) images / uniform labels (SURVEY.md §8d synthetic inputs), generated once, resident in HBM."""

    def __init__(self, n, batch, device, rank=0, world=1, seed=3, side=32, classes=10):
        g = torch.Generator().manual_seed(seed + 1000 * rank)
        per = batch // world
        self.n_batches = max(n // batch, 1)
        self.x = [torch.rand(per, 3, side, side, generator=g).to(device) for _ in range(min(self.n_batches, 8))]
        self.y = [torch.randint(0, classes, (per,), generator=g).to(device) for _ in range(min(self.n_batches, 8))]

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for b in range(self.n_batches):
            yield self.x[b % len(self.x)], self.y[b % len(self.y)]


# ----------------------------------------------------------------------------------------------- loops
def accuracy(output, target):
    return (output.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])


def train(train_loader, trainer, optimizer, epoch, args, log):
    """main_perturb.py:153-225.  Device-side accumulation; one read-back per print_freq iterations."""
    losses, top1 = AverageMeter(), AverageMeter()
    trainer.model.train()
    wp_steps = len(train_loader)
    norm_l2, norm_linf, pending = [], [], []

    def flush():
        # (the host reads results here anyway: every step issued so far is verified against a given-up grid barrier of the in-launch
        # BatchNorm, and run again on the two-launch forms if one did — train_step.AfanTrainer.flush_guard, grid_guard.py)
        if trainer.flush_guard():
            log("in-launch BatchNorm: a grid barrier gave up; the affected steps were run again on the two-launch forms "
                "(their logged loss / accuracy values are invalid)")
        for loss_t, prec_t, n in pending:
            losses.update(loss_t.item(), n)
            top1.update(prec_t.item(), n)
        pending.clear()

    for i, (inp, target) in enumerate(train_loader):
        if args.max_iters and i >= args.max_iters:
            break
        if epoch == 0:
            train_step.warmup_lr(i, optimizer, warm_up_steps=wp_steps, max_lr=args.lr)
        r = trainer.step(inp, target)
        norm_l2.append(r["l2"])
        norm_linf.append(r["linf"])
        pending.append((r["loss"], r["prec1"], inp.size(0)))
        if i % args.print_freq == 0:
            flush()
            log("Epoch: [{0}][{1}/{2}]\t"
                "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                "Accuracy {top1.val:.3f} ({top1.avg:.3f})\t".format(epoch, i, len(train_loader), loss=losses, top1=top1))
    flush()
    norm_mean_l2 = torch.mean(torch.cat(norm_l2, dim=0)).cpu()
    norm_mean_linf = torch.mean(torch.cat(norm_linf, dim=0)).cpu()
    log("l2 mean = {}".format(norm_mean_l2))
    log("linf mean = {}".format(norm_mean_linf))
    log("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg, losses.avg, norm_mean_l2.numpy(), norm_mean_linf.numpy()


def validate(val_loader, model, criterion, args, log):
    """main_perturb.py:227-263"""
    losses, top1 = AverageMeter(), AverageMeter()
    model.eval()
    for i, (inp, target) in enumerate(val_loader):
        with torch.no_grad():
            output = model(inp, end_point=model.layer_number, start_point=0)
            loss = criterion(output, target)
        losses.update(loss.float().item(), inp.size(0))
        top1.update(accuracy(output.float(), target).item(), inp.size(0))
        if i % args.print_freq == 0:
            log("Test: [{0}/{1}]\t"
                "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                "Accuracy {top1.val:.3f} ({top1.avg:.3f})".format(i, len(val_loader), loss=losses, top1=top1))
    log("valid_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg, losses.avg


def main(argv=None):
    args = parser.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    placement = host.place_rank(local)         # this rank's threads on one block of cores of its GPU's NUMA node (before the GPU is touched)
    if not torch.cuda.is_available():
        raise RuntimeError("main_perturb.py needs an MI355X: this build has no CPU path (oracle/ is test infrastructure)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    def log(*a):
        if rank == 0:
            print(*a, flush=True)

    log(args)
    log("host placement:", {k: v for k, v in placement.items() if k != "restore"})
    if args.seed:
        setup_seed(args.seed)
    if args.arch == "resnet50" and not args.synthetic:
        raise SystemExit("--arch resnet50 is the ImageNet-shape synthetic configuration: pass --synthetic N")
    ctor, _ = resnet_s.ARCHS[args.arch]
    model = ctor()                      # constructed after seeding, on the host generator, like main_perturb.py:64
    layer_number = model.layer_number
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model.set_channels_last(args.layout == "nhwc").to(dev)
    criterion = nn.CrossEntropyLoss()
    trainer = train_step.AfanTrainer(model, criterion, steps=args.steps, gamma=args.gamma, eps=args.eps,
                                     perturb_idx=args.perturb_idx, layer_number=layer_number, randinit=args.randinit,
                                     clip=args.clip, lr=args.lr, momentum=args.momentum,
                                     weight_decay=args.weight_decay, dual_bn=args.dual_bn)
    optimizer = trainer.optimizer
    vendor = resnet_s.general_convs(model)
    log("convolutions outside the library's kernels: {}{}".format(
        len(vendor), " (general f32-MFMA kernels; --dtype bf16 --layout nhwc is the tuned bf16 MFMA path)" if vendor else ""))
    decreasing_lr = list(map(int, args.decreasing_lr.split(",")))
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=decreasing_lr, gamma=0.1)

    if args.synthetic:
        side, classes = (224, 1000) if args.arch == "resnet50" else (32, 10)
        train_loader = SyntheticLoader(args.synthetic, args.batch_size, dev, rank, world, side=side, classes=classes)
        val_loader = test_loader = SyntheticLoader(max(args.synthetic // 10, args.batch_size), args.batch_size, dev,
                                                   side=side, classes=classes)
    else:
        tr, va, te = _load_cifar10(args.data)
        shared = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)     # rank 0's draw (seeded or not) for everyone
        if world > 1:
            shared = shared.to(dev)
            dist.broadcast(shared, src=0)
        train_loader = DeviceLoader(tr[0], tr[1], args.batch_size, dev, True, rank, world, seed=int(shared.item()))
        val_loader = DeviceLoader(va[0], va[1], args.batch_size, dev, False, drop_last=False)
        test_loader = DeviceLoader(te[0], te[1], args.batch_size, dev, False, drop_last=False)

    best_prec1, start_epoch = 0, 0
    if args.resume:
        log("resume from checkpoint")
        ck = torch.load(os.path.join(args.save_dir, "checkpoint.pt"), map_location=dev)
        best_prec1, start_epoch = ck["best_prec1"], ck["epoch"]
        model.load_state_dict(ck["state_dict"])
        trainer.arena.refresh_shadow()
        optimizer.load_state_dict(ck["optimizer"])
        scheduler.load_state_dict(ck["scheduler"])

    all_result, train_acc, ta, test_ta = {}, [], [], []
    os.makedirs(args.save_dir, exist_ok=True)
    all_norm_result = {"l2": {}, "linf": {}}
    for epoch in range(start_epoch, args.epochs):
        log(optimizer.state_dict()["param_groups"][0]["lr"])
        acc, _, n2, ninf = train(train_loader, trainer, optimizer, epoch, args, log)
        all_norm_result["l2"][epoch + 1] = n2
        all_norm_result["linf"][epoch + 1] = ninf
        tacc, _ = validate(val_loader, model, criterion, args, log)
        test_tacc, _ = validate(test_loader, model, criterion, args, log)
        scheduler.step()
        train_acc.append(acc), ta.append(tacc), test_ta.append(test_tacc)
        is_best = tacc > best_prec1
        best_prec1 = max(tacc, best_prec1)
        if rank == 0:
            state = {"epoch": epoch + 1, "state_dict": model.state_dict(), "best_prec1": best_prec1,
                     "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict()}
            if is_best:
                torch.save(state, os.path.join(args.save_dir, "best_model.pt"))
            torch.save(state, os.path.join(args.save_dir, "checkpoint.pt"))
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                plt.plot(train_acc, label="train_acc"), plt.plot(ta, label="TA"), plt.plot(test_ta, label="test_TA")
                plt.legend()
                plt.savefig(os.path.join(args.save_dir, "net_train.png"))
                plt.close()
            except ImportError:
                pass
            all_result.update(train=train_acc, test_ta=test_ta, ta=ta)
            pickle.dump(all_result, open(os.path.join(args.save_dir, "result.pkl"), "wb"))
            pickle.dump(all_norm_result, open(os.path.join(args.save_dir, "result_norm.pkl"), "wb"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
