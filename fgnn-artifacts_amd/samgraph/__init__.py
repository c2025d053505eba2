"""MI355X-native drop-in for the reference's `samgraph` Python package (samgraph/__init__.py)."""
