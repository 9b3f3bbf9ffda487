"""Evaluation helpers next to the hot path."""
