"""x2-vlm_amd: MI355X-native X^2-VLM pre-training step (HIP kernels behind a C ABI + host mirror of
the reference's module API).  Import with importlib.import_module("x2-vlm_amd") or `import x2vlm_amd`."""
