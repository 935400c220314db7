"""Import stub (test infrastructure only): the reference imports torchvision solely for
`normalize` (unidepth/models/unidepthv2/unidepthv2.py:14,289). Restated below."""
