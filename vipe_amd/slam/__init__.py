"""Host-side mirror of the reference's vipe.slam operator API for the update-iteration hot path."""
