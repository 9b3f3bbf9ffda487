"""Test infrastructure only: CPU oracle for the hot path (see ref_cpu.py header). Never imported by qllm_amd/."""
