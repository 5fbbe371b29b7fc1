"""MI355X-native batched tabular-MDP engine with Colosseum's BaseMDP / dynamic-programming call surface."""
__version__ = "0.1.0"
