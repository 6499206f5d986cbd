"""HIP sources of liblidf_hip.so and their build script (build.py)."""
