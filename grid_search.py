"""Grid search (/root/reference/grid_search.py:138-175).  Under
`python -m torch.distributed.run --nproc-per-node 8 grid_search.py` the configurations run
one per MI355X, no collectives (gat_amd/trials.py)."""
import itertools

import torch

from gat_amd.trials import Config, best_of, run_trials, world, write_results


def main():
    grid = dict(optimizer=[torch.optim.AdamW], lr=[0.04], weight_decay=[5e-4], epochs=[30],
                patience=[3], hidden_dim=[32], dropout=[0.4])
    configs = [Config(*values) for values in itertools.product(*grid.values())]
    results = run_trials(configs)
    if world()[0] == 0:
        best, score = best_of(results)
        print(f"Best config: {best} with composite score: {score}")
        write_results(results, "./results/all_results_grid_search.txt")
    return results


if __name__ == "__main__":
    main()
