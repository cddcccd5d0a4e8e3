"""qwen3tts -- host-side mirror of AtomGradient/swift-qwen3-tts' public surface over the
MI355X-native HIP engine (libq3tts_hip.so, C ABI in include/q3tts.h)."""
from .model import (AudioGenerationInfo, GenerationRequest, GenerationResult, Qwen3TTSError, Qwen3TTSModel,
                    chat_template_ids)

__all__ = ["AudioGenerationInfo", "GenerationRequest", "GenerationResult", "Qwen3TTSError", "Qwen3TTSModel",
           "chat_template_ids"]
