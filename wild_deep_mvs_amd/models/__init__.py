"""Mirror of the reference's ``models`` package tree (same import paths below ``models.``, same class,
function and parameter names) with the depth-inference hot path running on the pscv HIP engine.

``wild_deep_mvs_amd.install_as_models()`` aliases this package as top-level ``models`` so that the
reference's ``train.py`` / ``depthmap_eval.py`` / ``evaluation/pipeline_utils.py`` import it unchanged."""
