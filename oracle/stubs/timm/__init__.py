"""Import stub (test infrastructure only) for the un-installed `timm` package. Only what the
UniDepthV2 ViT import chain touches is provided; ConvNeXt helpers raise if instantiated."""
