"""Single training run with the reference's fixed hyper-parameters
(/root/reference/run.py:14-21, 56-60) on the MI355X GAT layer."""
import os

import numpy as np
import torch

from gat_amd import trials


class Config:
    optimizer = torch.optim.AdamW
    lr = 0.0425
    weight_decay = 0.0004807430799298252
    epochs = 30
    patience = 5
    hidden_dim = 30
    dropout = 0.425


def main():
    os.makedirs("./results", exist_ok=True)
    print("Start preprocessing / splitting / converting to graph...")
    train_data, test_data, y_train, _ = trials.split_data()
    print(f"Graphs ready: {train_data.num_nodes} train nodes, {test_data.num_nodes} test nodes, "
          f"{train_data.num_features} features, {len(np.unique(y_train))} classes")
    model = trials.initialize_model(train_data, y_train, Config())
    model.checkpoint_path = "best_model.pth"
    return model.train_model(train_data, test_data)


if __name__ == "__main__":
    main()
