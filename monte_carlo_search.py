"""Monte-Carlo search (/root/reference/monte_carlo_search.py:18, 113-178): NUM_SAMPLES
configurations drawn uniformly from the reference's ranges -- learning rate, weight decay and
dropout from continuous intervals, the rest from lists -- NUM_RUNS run each.  The reference
draws one configuration, trains it, draws the next; the draws do not depend on the results, so
they are all made up front (identically on every rank, from `--seed`; the reference draws
unseeded) and handed out one trial per GPU by gat_amd/trials.py's dynamic queue."""
import argparse
import random

import torch

from gat_amd.trials import Config, best_of, run_experiment, run_trials, world, write_results

NUM_RUNS = 1        # monte_carlo_search.py:17
NUM_SAMPLES = 50    # monte_carlo_search.py:18

RANGES = {          # monte_carlo_search.py:154-162
    "optimizers": [torch.optim.AdamW],
    "lr": (0.03, 0.05),
    "weight_decay": (4.5e-4, 6e-4),
    "epochs": [25, 35],
    "patience": [3, 7],
    "hidden_dim": [24, 40],
    "dropout": (0.4, 0.45),
}


def sample_configs(num_samples, seed, ranges=RANGES):
    rng = random.Random(seed)
    return [Config(optimizer=rng.choice(ranges["optimizers"]), lr=rng.uniform(*ranges["lr"]),
                   weight_decay=rng.uniform(*ranges["weight_decay"]), epochs=rng.choice(ranges["epochs"]),
                   patience=rng.choice(ranges["patience"]), hidden_dim=rng.choice(ranges["hidden_dim"]),
                   dropout=rng.uniform(*ranges["dropout"])) for _ in range(num_samples)]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=NUM_SAMPLES)
    ap.add_argument("--runs", type=int, default=NUM_RUNS)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--schedule", default="dynamic", choices=["static", "dynamic"])
    args = ap.parse_args(argv)
    configs = sample_configs(args.samples, args.seed)
    results = run_trials(configs, lambda c: run_experiment(c, args.runs), schedule=args.schedule)
    if world()[0] == 0:
        best, score = best_of(results)
        print(f"Best config: {best} with composite score: {score}")
        write_results(results, "./results/all_results_monte_carlo.txt")
    return results


if __name__ == "__main__":
    main()
