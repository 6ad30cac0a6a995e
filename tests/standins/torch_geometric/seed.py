"""torch_geometric.seed.seed_everything (100M/nb-sample.py:16,73).  Stand-in, tests only."""
import random

import numpy as np
import torch


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
