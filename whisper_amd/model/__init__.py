from .load import load_onnx_models, get_tokenizer_for_model, resolve_weights  # noqa: F401
