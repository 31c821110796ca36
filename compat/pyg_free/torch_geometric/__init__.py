"""Optional second ``PYTHONPATH`` entry for machines WITHOUT torch-geometric (this image): the three names the
reference's drivers take from it -- ``torch_geometric.data.Data`` / ``Batch`` (train.py:247, validation.py:56) and
``torch_geometric.loader.DataLoader`` (train.py:6) -- backed by the engine's own containers.  Never put this
directory on the path where the real package is installed: it would shadow it."""
from . import data, loader  # noqa: F401
