from .model_Uni import Uni_model  # noqa: F401
