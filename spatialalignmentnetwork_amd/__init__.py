"""MI355X-native reconstruction + alignment hot path."""
