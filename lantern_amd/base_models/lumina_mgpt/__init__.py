from .eagle_inference_solver import FlexARInferenceSolver  # noqa: F401
