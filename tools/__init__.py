"""Measurement tools around bench.py (probes, sweeps, profile condensers); not part of the product package."""
