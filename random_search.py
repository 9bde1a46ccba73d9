"""Random search (/root/reference/random_search.py:17-18, 113-180): NUM_SAMPLES draws from
the reference's ranges, NUM_RUNS runs each.  The draws are made up front (identically on
every rank, from `--seed`; the reference draws unseeded) and then sharded one trial per
GPU by gat_amd/trials.py."""
import argparse
import random

import torch

from gat_amd.trials import Config, best_of, run_experiment, run_trials, world, write_results

NUM_SAMPLES = 10
NUM_RUNS = 3

RANGES = {
    "optimizers": [torch.optim.AdamW],
    "lr": [0.0325, 0.035, 0.0375, 0.04, 0.0425],
    "weight_decay": (0.00045, 0.00055),
    "epochs": [30, 40],
    "patience": [5],
    "hidden_dim": [28, 30, 32, 34],
    "dropout": [0.35, 0.375, 0.4, 0.425, 0.45],
}


def sample_configs(num_samples, seed):
    rng = random.Random(seed)
    return [Config(optimizer=rng.choice(RANGES["optimizers"]), lr=rng.choice(RANGES["lr"]),
                   weight_decay=rng.uniform(*RANGES["weight_decay"]), epochs=rng.choice(RANGES["epochs"]),
                   patience=rng.choice(RANGES["patience"]), hidden_dim=rng.choice(RANGES["hidden_dim"]),
                   dropout=rng.choice(RANGES["dropout"])) for _ in range(num_samples)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=NUM_SAMPLES)
    ap.add_argument("--runs", type=int, default=NUM_RUNS)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--schedule", default="dynamic", choices=["static", "dynamic"],
                    help="how trials are handed to the GPUs (gat_amd/trials.py)")
    args = ap.parse_args()
    configs = sample_configs(args.samples, args.seed)
    results = run_trials(configs, lambda c: run_experiment(c, args.runs), schedule=args.schedule)
    if world()[0] == 0:
        best, score = best_of(results)
        print(f"Best config: {best} with composite score: {score}")
        write_results(results, "./results/all_results_random_search.txt")
    return results


if __name__ == "__main__":
    main()
