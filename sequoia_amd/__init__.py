"""MI355X-native hot path of Sequoia tree speculative decoding."""
