"""MI355X-native speech features backend (see README.md)"""
