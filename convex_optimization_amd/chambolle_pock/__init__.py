from .algorithm import chambolle_pock

__all__ = ["chambolle_pock"]
