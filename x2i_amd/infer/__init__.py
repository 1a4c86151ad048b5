"""Sampling harness: this build's counterpart of the reference's infer/inference_*.py (Row H)."""
