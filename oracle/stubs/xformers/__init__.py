"""TEST INFRASTRUCTURE ONLY -- stand-in for the un-vendored xformers package so that the reference's UniDepthV1 decoder can be
EXECUTED here.  Only xformers.components.attention.NystromAttention exists; `xformers.ops` deliberately does not, so the reference's
DINOv2 code keeps taking its F.scaled_dot_product_attention fallback (metadinov2/attention.py:20-28)."""
