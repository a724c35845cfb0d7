"""MI355X-native GRL hot path."""
