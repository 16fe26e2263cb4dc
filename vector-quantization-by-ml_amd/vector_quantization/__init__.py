"""MI355X-native drop-in for the nearest-codebook hot path of MisterBourbaki/vector-quantization-by-ml."""
