"""Test-infrastructure package: CPU restatements of the reference (see the module docstrings)."""
