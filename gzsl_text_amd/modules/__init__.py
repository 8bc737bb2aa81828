"""Drop-in mirrors of the reference's `modules` package (icd_modules, graph_module, generative_module,
common_module, decoder_module, lm_module): same class names, constructor signatures, attribute names and
state-dict keys; forwards run on the HIP kernels."""
