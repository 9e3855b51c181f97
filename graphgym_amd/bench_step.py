"""bench.py --mode step (filled in below: the data-parallel ID-GCN training step with the gradient all-reduce
inside the timed region)."""


def run(args, rank, world, dev):
    raise SystemExit("--mode step: not built yet")
