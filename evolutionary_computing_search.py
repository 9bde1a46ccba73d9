"""Evolutionary search (/root/reference/evolutionary_computing_search.py:18-20, 115-222).

The outer loop over generations is sequential (each generation breeds from the scores of the
one before); the evaluations INSIDE a generation are independent, so each population is handed
to gat_amd/trials.py and runs one trial per GPU.  Selection, crossover
and mutation follow the reference: the better half of POPULATION_SIZE survives, every survivor
adds two children bred with a random survivor, each child field mutates with MUTATION_RATE.
Every rank draws from the same seeded generator and sees the same gathered scores, so all ranks
breed the same next generation without exchanging it.

One thing is not repeated: the reference trains every population twice -- once at the end of a
generation to track the best configuration (:196-200), and again at the start of the next one to
rank the same configurations (:170-172).  Here a population is trained once and its scores
serve both purposes."""
import argparse
import copy
import os
import random

import torch

from gat_amd.trials import Config, run_experiment, run_trials, world

NUM_RUNS = 1            # evolutionary_computing_search.py:17
POPULATION_SIZE = 10    # :18
NUM_GENERATIONS = 5     # :19
MUTATION_RATE = 0.1     # :20

RANGES = {              # :208-216
    "optimizers": [torch.optim.AdamW],
    "lr": [0.039, 0.04, 0.041, 0.042, 0.0425],
    "weight_decay": (0.00046, 0.00052),
    "epochs": [30, 40],
    "patience": [5],
    "hidden_dim": [28, 30, 32, 34, 36],
    "dropout": [0.35, 0.375, 0.4, 0.425, 0.45],
}
FIELDS = ("optimizer", "lr", "weight_decay", "epochs", "patience", "hidden_dim", "dropout")


def create_initial_population(size, ranges, rng):
    return [Config(optimizer=rng.choice(ranges["optimizers"]), lr=rng.choice(ranges["lr"]),
                   weight_decay=rng.uniform(*ranges["weight_decay"]), epochs=rng.choice(ranges["epochs"]),
                   patience=rng.choice(ranges["patience"]), hidden_dim=rng.choice(ranges["hidden_dim"]),
                   dropout=rng.choice(ranges["dropout"])) for _ in range(size)]


def mutate_config(config, ranges, rng, rate=MUTATION_RATE):
    """each field but the optimizer is redrawn with probability `rate` (:130-143)"""
    for name in FIELDS[1:]:
        if rng.random() < rate:
            r = ranges[name]
            setattr(config, name, rng.uniform(*r) if isinstance(r, tuple) else rng.choice(r))
    return config


def crossover_configs(a, b, rng):
    """two children, every field from one parent or the other (:145-164)"""
    pick = lambda: Config(**{name: rng.choice([getattr(a, name), getattr(b, name)]) for name in FIELDS})
    return pick(), pick()


def next_population(scored, ranges, rng, population_size=POPULATION_SIZE):
    """scored: [(composite_score, config)] of one generation -> the next one (:174-183)"""
    ranked = sorted(scored, key=lambda t: t[0], reverse=True)    # stable, as list.sort is
    top = ranked[:population_size // 2]
    out = []
    for _, config in top:
        out.append(config)
        c1, c2 = crossover_configs(config, rng.choice(top)[1], rng)
        out.extend([mutate_config(c1, ranges, rng), mutate_config(c2, ranges, rng)])
    return out


def evolutionary_search(ranges, evaluate, seed=0, population_size=POPULATION_SIZE,
                        generations=NUM_GENERATIONS):
    """evaluate(list of Config) -> list of composite scores, same order (all ranks alike)."""
    rng = random.Random(seed)
    population = create_initial_population(population_size, ranges, rng)
    scores = evaluate(population)
    best_config, best_score = None, 0
    for generation in range(generations):
        if world()[0] == 0:
            print(f"Generation {generation + 1}")
        population = next_population(list(zip(scores, population)), ranges, rng, population_size)
        scores = evaluate(population)
        for score, config in zip(scores, population):
            if score > best_score:
                best_score, best_config = score, copy.copy(config)
    return best_config, best_score


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--population", type=int, default=POPULATION_SIZE)
    ap.add_argument("--generations", type=int, default=NUM_GENERATIONS)
    ap.add_argument("--runs", type=int, default=NUM_RUNS)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    def evaluate(population):
        results = run_trials(population, lambda c: run_experiment(c, args.runs), schedule="dynamic")
        return [r.composite_score for r in results]

    best, score = evolutionary_search(RANGES, evaluate, args.seed, args.population, args.generations)
    if world()[0] == 0:
        print(f"Best config: {best} with composite score: {score}")
        os.makedirs("./results", exist_ok=True)
        with open("./results/best_result_evolutionary.txt", "w") as f:
            f.write(f"Best Config: {best}\nComposite Score: {score}\n")
    return best, score


if __name__ == "__main__":
    main()
