"""Learnable multi-layer A-FAN entry point — flags, stdout lines and output files of the reference's
Classification/main_learnable.py (flags :28-56, loop :110-170, train :175-277, sum_project :369-378); single GPU, like
the reference.  Additions (optional): --dtype, --synthetic, --max_iters.  The iteration body is
learnable.LearnableTrainer.step; data loading / validation / checkpoint layout are shared with main_perturb.py."""
import argparse
import os
import pickle
import sys

import torch
import torch.nn as nn

if __package__ in (None, ""):  # executed as a script: import the hyphenated package by path
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module("cv_a-fan_amd")
    resnet_s, learnable, host, mp = _pkg.resnet_s, _pkg.learnable, _pkg.host, importlib.import_module("cv_a-fan_amd.main_perturb")
else:
    from . import host, resnet_s, learnable
    from . import main_perturb as mp

parser = argparse.ArgumentParser(description="Learnable multi-layer A-FAN CIFAR-10 training on MI355X")
# ---- base setting (main_learnable.py:28-35)
parser.add_argument("--data", type=str, default="../data", help="location of the data corpus")
parser.add_argument("--print_freq", default=50, type=int, help="print frequency")
parser.add_argument("--seed", default=None, type=int, help="random seed")
parser.add_argument("--gpu", type=int, default=0, help="gpu device id")
parser.add_argument("--resume", action="store_true", help="resume from checkpoint")
parser.add_argument("--save_dir", help="The directory used to save the trained models", default="res56s_aug_learnable", type=str)
# ---- optimizer setting (:38-44)
parser.add_argument("--batch_size", type=int, default=128, help="batch size")
parser.add_argument("--lr", default=0.1, type=float, help="initial learning rate")
parser.add_argument("--momentum", default=0.9, type=float, help="momentum")
parser.add_argument("--weight_decay", default=5e-4, type=float, help="weight decay")
parser.add_argument("--epochs", default=200, type=int, help="number of total epochs to run")
parser.add_argument("--decreasing_lr", default="50,150", help="decreasing strategy")
# ---- A-FAN setting (:47-56)
parser.add_argument("--steps", default=3, type=int, help="PGD-steps")
parser.add_argument("--gamma", help="index of PGD gamma", default=1, type=float)
parser.add_argument("--eps", default=2, type=float)
parser.add_argument("--randinit", action="store_true", help="whether using randinit")
parser.add_argument("--clip", action="store_true", help="whether using clip")
parser.add_argument("--w_lr", default=0.01, type=float, help="learning rate of the mixing weights")
parser.add_argument("--init_weight", default=(1 / 9), type=float, help="initial weight for ETA")
parser.add_argument("--l1_coef", default=1, type=float, help="coefficient of the L1 penalty on the mixing weights")
# ---- additions
parser.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="backbone compute dtype")
parser.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"],
                    help="internal activation / weight layout (nhwc: the library's MFMA convolutions; nchw: the general fp32-arithmetic kernels)")
parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images instead of CIFAR-10")
parser.add_argument("--max_iters", type=int, default=0, help="stop each epoch after this many iterations (0 = all)")


def train(train_loader, trainer, epoch, args, log):
    """main_learnable.py:175-277; metrics are read back every --print_freq iterations instead of every iteration."""
    losses, top1 = mp.AverageMeter(), mp.AverageMeter()
    trainer.model.train()
    wp_steps = len(train_loader)
    norm_l2, norm_linf = [], []
    pend = []            # (loss, prec1, n) device scalars of EVERY iteration, flushed into the meters at print time
    for i, (inp, target) in enumerate(train_loader):
        if args.max_iters and i >= args.max_iters:
            break
        if epoch == 0:                                              # warmup_lr (:340-345)
            lr = min(i * args.lr / (wp_steps - 1), args.lr) if wp_steps > 1 else args.lr
            for g in trainer.optimizer.param_groups:
                g["lr"] = lr
        r = trainer.step(inp, target)
        norm_l2.append(r["l2"])
        norm_linf.append(r["linf"])
        pend.append((r["loss"].float(), r["prec1"], inp.size(0)))    # (graph replays hand out copies of the small outputs)
        if i % args.print_freq == 0:
            if trainer.flush_guard():    # (grid_guard.py: steps that ran on a given-up grid barrier's partial totals were run again)
                log("in-launch BatchNorm: a grid barrier gave up; the affected steps were run again on the two-launch forms")
            for lo, pr, n in pend:       # one host sync per print, every iteration counted (:263-268 update per iteration)
                losses.update(lo.item(), n)
                top1.update(pr.item(), n)
            pend = []
            log("Epoch: [{0}][{1}/{2}]\t"
                "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                "Accuracy {top1.val:.3f} ({top1.avg:.3f})\t".format(epoch, i, len(train_loader), loss=losses, top1=top1))
    trainer.flush_guard()
    for lo, pr, n in pend:
        losses.update(lo.item(), n)
        top1.update(pr.item(), n)
    l2m = torch.cat(norm_l2, dim=1).mean(dim=1).cpu()
    linfm = torch.cat(norm_linf, dim=1).mean(dim=1).cpu()
    log("l2 mean = {}".format(l2m))
    log("linf mean = {}".format(linfm))
    log("train_accuracy {top1.avg:.3f}".format(top1=top1))
    return top1.avg, losses.avg, l2m.numpy(), linfm.numpy()


def main(argv=None):
    args = parser.parse_args(argv)
    host.place_rank(int(args.gpu))
    if not torch.cuda.is_available():
        raise RuntimeError("main_learnable.py needs an MI355X: this build has no CPU path (oracle/ is test infrastructure)")
    torch.cuda.set_device(int(args.gpu))
    dev = torch.device("cuda", int(args.gpu))
    log = lambda *a: print(*a, flush=True)
    log(args)
    if args.seed:
        mp.setup_seed(args.seed)
    model = resnet_s.resnet56(init_weight_eta=args.init_weight)       # :73
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model.set_channels_last(args.layout == "nhwc").to(dev)
    criterion = nn.CrossEntropyLoss()
    trainer = learnable.LearnableTrainer(model, criterion, steps=args.steps, gamma=args.gamma, eps=args.eps,
                                         randinit=args.randinit, clip=args.clip, lr=args.lr, w_lr=args.w_lr,
                                         l1_coef=args.l1_coef, momentum=args.momentum, weight_decay=args.weight_decay)
    optimizer, optimizer_w = trainer.optimizer, trainer.optimizer_w
    vendor = resnet_s.general_convs(model)
    log("convolutions outside the library's kernels: {}{}".format(
        len(vendor), " (general f32-MFMA kernels; --dtype bf16 --layout nhwc is the tuned bf16 MFMA path)" if vendor else ""))
    decreasing_lr = list(map(int, args.decreasing_lr.split(",")))
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=decreasing_lr, gamma=0.1)
    if args.synthetic:
        train_loader = mp.SyntheticLoader(args.synthetic, args.batch_size, dev)
        val_loader = test_loader = mp.SyntheticLoader(max(args.synthetic // 10, args.batch_size), args.batch_size, dev)
    else:
        tr, va, te = mp._load_cifar10(args.data)
        train_loader = mp.DeviceLoader(tr[0], tr[1], args.batch_size, dev, True)
        val_loader = mp.DeviceLoader(va[0], va[1], args.batch_size, dev, False, drop_last=False)
        test_loader = mp.DeviceLoader(te[0], te[1], args.batch_size, dev, False, drop_last=False)

    best_prec1, start_epoch = 0, 0
    if args.resume:
        log("resume from checkpoint")
        ck = torch.load(os.path.join(args.save_dir, "checkpoint.pt"), map_location=dev)
        best_prec1, start_epoch = ck["best_prec1"], ck["epoch"]
        model.load_state_dict(ck["state_dict"])
        trainer.arena.refresh_shadow()
        optimizer.load_state_dict(ck["optimizer"])
        optimizer_w.load_state_dict(ck["optimizer_w"])
        scheduler.load_state_dict(ck["scheduler"])

    all_result, train_acc, ta, test_ta = {}, [], [], []
    all_norm_result = {"l2": {}, "linf": {}}
    os.makedirs(args.save_dir, exist_ok=True)
    for epoch in range(start_epoch, args.epochs):
        for num in range(9):
            log("weight" + str(num + 1) + " = ", model.w[num].item())
        log(optimizer.state_dict()["param_groups"][0]["lr"])
        log(optimizer_w.state_dict()["param_groups"][0]["lr"])
        acc, _, n2, ninf = train(train_loader, trainer, epoch, args, log)
        all_norm_result["l2"][epoch + 1] = n2
        all_norm_result["linf"][epoch + 1] = ninf
        tacc, _ = mp.validate(val_loader, model, criterion, args, log)
        test_tacc, _ = mp.validate(test_loader, model, criterion, args, log)
        scheduler.step()
        train_acc.append(acc), ta.append(tacc), test_ta.append(test_tacc)
        is_best = tacc > best_prec1
        best_prec1 = max(tacc, best_prec1)
        state = {"epoch": epoch + 1, "state_dict": model.state_dict(), "best_prec1": best_prec1,
                 "optimizer": optimizer.state_dict(), "optimizer_w": optimizer_w.state_dict(),
                 "scheduler": scheduler.state_dict()}
        if is_best:
            torch.save(state, os.path.join(args.save_dir, "best_model.pt"))
        torch.save(state, os.path.join(args.save_dir, "checkpoint.pt"))
        all_result.update(train=train_acc, test_ta=test_ta, ta=ta)
        pickle.dump(all_result, open(os.path.join(args.save_dir, "result.pkl"), "wb"))
        pickle.dump(all_norm_result, open(os.path.join(args.save_dir, "result_norm.pkl"), "wb"))


if __name__ == "__main__":
    main()
