"""Import stub (test infrastructure only): reference imports wandb at unidepth/utils/visualization.py:11 without using it on the infer() path."""
