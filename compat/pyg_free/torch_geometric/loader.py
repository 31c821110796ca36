"""``torch_geometric.loader.DataLoader`` as train.py:150-162 uses it: over a dataset of nested dicts of tensors it
behaves like ``torch.utils.data.DataLoader`` with the default collate."""
from torch.utils.data import DataLoader  # noqa: F401
