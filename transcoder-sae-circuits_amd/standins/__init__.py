"""Stand-ins for what cannot be fetched offline -- NOT part of the accelerated path.

The reference gets GPT-2 small from TransformerLens, its 24 dictionaries from the HF hub and its prompts from
OpenWebText / the IOI templates.  None of these exist in the build image or on the GPU box, so tests, golden
fixtures and benchmarks use the seeded substitutes in this sub-package: ``synthetic_model`` (a GPT-2-small-shaped
``HookedTransformer`` look-alike), ``synthetic_dicts`` (random-init dictionaries with the reference's init
statistics) and ``synthetic_prompts`` (IOI-shaped prompt sets).  The product modules never import them except
``load_model_and_encoders.install_synthetic``, the documented way to seed the model/encoder cache offline."""
