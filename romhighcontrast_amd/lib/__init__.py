"""Host-side mirror of the reference's ``src/lib`` package (same module and class names)."""
