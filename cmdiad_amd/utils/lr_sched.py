"""Learning-rate schedule of the distillation trainer (reference utils/lr_sched.py:4-17):
linear warm-up over ``args.warmup_epochs`` (fractional epochs), then constant ``args.lr``."""


def adjust_learning_rate(optimizer, epoch, args):
    lr = args.lr * epoch / args.warmup_epochs if epoch < args.warmup_epochs else args.lr
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
