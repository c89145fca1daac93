"""Host side of the C-ABI boundary (include/mqe_hip.h): ctypes mirror of the descriptor, descriptor builder from
an MQE config class, and the loader of the HIP engine library."""
