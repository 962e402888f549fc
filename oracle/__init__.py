"""CPU oracle for the guided banded alignment path (test infrastructure only)."""
