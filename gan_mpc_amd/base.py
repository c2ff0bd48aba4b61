"""The model protocol the policies are written against -- the drop-in surface a user of the reference
knows (reference base.py:4-49: BaseCostModel / BaseDynamicsModel / BaseCriticModel and the NN mix-ins).

The names and call signatures are the contract; the implementation here is a small declarative layer:
each protocol class lists its required methods once and `_Protocol` turns a missing override into the
same NotImplementedError the reference raises, with a message that names the class and the method."""


def _abstract(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} must implement {name}()")

    method.__name__ = name
    method.__isabstract__ = True
    return method


class _Protocol:
    """Base of the protocol classes: REQUIRED names the methods a concrete model has to provide."""

    REQUIRED = ()

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        for name in cls.__dict__.get("REQUIRED", ()):
            if name not in cls.__dict__:
                setattr(cls, name, _abstract(name))

    @classmethod
    def missing(cls):
        """Names of the protocol methods this class still has to override."""
        return [n for klass in cls.__mro__ for n in klass.__dict__.get("REQUIRED", ())
                if getattr(getattr(cls, n, None), "__isabstract__", False)]


class _ConfiguredModel(_Protocol):
    def __init__(self, config):
        self.config = config


class BaseCostModel(_ConfiguredModel):
    """init(*args) -> params;  get_cost(x, u, t, *cost_args) -> scalar"""
    REQUIRED = ("init", "get_cost")


class BaseDynamicsModel(_ConfiguredModel):
    """init(*args) -> params;  predict(x, u, t, *dynamics_args) -> next state (+ carry)"""
    REQUIRED = ("init", "predict")


class BaseCriticModel(_ConfiguredModel):
    """init(*args) -> params;  predict(xseq, *args) -> score"""
    REQUIRED = ("init", "predict")


class BaseNN(_Protocol):
    """get_init_params(*args) -> the arguments of the network's init"""
    REQUIRED = ("get_init_params",)


class BaseCostNN(BaseNN):
    REQUIRED = ("get_cost",)


class BaseDynamicsNN(BaseNN):
    REQUIRED = ("get_carry",)
