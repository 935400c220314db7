"""Import stub (test infrastructure only): reference imports cv2 at unidepth/utils/distributed.py:7 without using it on the infer() path."""
