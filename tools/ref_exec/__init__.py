"""Execute reference source files in this container under NumPy stand-ins (test-fixture generation only)."""
