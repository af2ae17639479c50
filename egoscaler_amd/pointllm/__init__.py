"""Host-side mirror of the reference's model API (egoscaler/models/pointllm/{builder,model_arch,
constant}.py): same names, argument meaning and error behaviour, compute on libegomi.so."""
from .model_arch import TrajPointLLMForCausalLM, PointLLMConfig, CausalLMOutput, GenerateOutput  # noqa: F401
from .builder import build_model, init_model, add_trajectory_token  # noqa: F401
