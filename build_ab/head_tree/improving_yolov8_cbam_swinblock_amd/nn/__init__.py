from .tasks import DetectionModel, parse_model, yaml_model_load  # noqa: F401
