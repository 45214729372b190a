"""Helpers of bench.py (the driver's contract lives in ../bench.py; these modules hold what its modes share)."""
