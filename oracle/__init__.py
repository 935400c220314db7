"""TEST INFRASTRUCTURE ONLY.  CPU oracle for the UniDepthV2 infer() path: `restate.py` (travels to the
GPU box), `ref_loader.py` (imports the real reference from /root/reference; authoring container only),
`synth.py` (seeded sensitised checkpoints), `make_golden.py` (fixtures under tests/golden/).
Nothing under unidepth_amd/ may import this package."""
