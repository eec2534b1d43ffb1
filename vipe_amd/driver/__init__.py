"""Clip-sharded multi-GPU driver: one process per GPU, independent clips, one result gather at the end."""
