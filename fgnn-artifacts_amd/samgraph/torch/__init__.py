"""`import samgraph.torch as sam` -- same entry point as the reference (samgraph/torch/__init__.py)."""
from samgraph.torch.adapter import *  # noqa: F401,F403
