"""ParameterClass: a table of per-Gaussian tensors, some of them trainable, together with an optimizer whose
per-row state follows the rows through pruning (`params[keep]`) and densification (`params.append_tensors(new)`).

Same constructor, methods and properties as the reference's optim/parameter_class.py:12-260 -- which builds on
tensordict; here the table is `TensorTable`, a plain name -> tensor mapping with a shared leading dimension, so
the class works with dicts, with this package's Gaussians3D / Gaussians2D records and with anything else that
has `.items()`.
"""
from __future__ import annotations

import copy
from typing import Callable, Dict, Iterable, Optional, Tuple, Union

import torch


class TensorTable(dict):
    """name -> tensor, all with the same number of rows.  Indexing by a string returns a column, by anything else a
    table of the selected rows."""

    def __init__(self, columns=()):
        super().__init__(columns.items() if hasattr(columns, "items") else columns)
        rows = {int(t.shape[0]) for t in self.values()}
        assert len(rows) <= 1, f"columns disagree on the number of rows: {sorted(rows)}"

    @property
    def batch_size(self) -> torch.Size:
        return torch.Size([next(iter(self.values())).shape[0]]) if self else torch.Size([0])

    shape = batch_size
    batch_dims = 1

    def __getitem__(self, key):
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        return TensorTable({name: t[key] for name, t in self.items()})

    def __getattr__(self, name):
        try:
            return dict.__getitem__(self, name)
        except KeyError:
            raise AttributeError(name) from None

    def apply(self, f: Callable[[torch.Tensor], torch.Tensor]) -> "TensorTable":
        return TensorTable({name: f(t) for name, t in self.items()})

    def to(self, *args, **kwargs) -> "TensorTable":
        return self.apply(lambda t: t.to(*args, **kwargs))

    def detach(self) -> "TensorTable":
        return self.apply(torch.Tensor.detach)

    def replace(self, **columns) -> "TensorTable":
        return TensorTable({**self, **columns})

    def to_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self)

    def new_zeros(self, rows: int) -> "TensorTable":
        return self.apply(lambda t: t.new_zeros((rows, *t.shape[1:])))

    @staticmethod
    def concat(first: "TensorTable", second: "TensorTable") -> "TensorTable":
        assert set(first) == set(second), f"{sorted(first)} != {sorted(second)}"
        return TensorTable({name: torch.cat([t, second[name]], dim=0) for name, t in first.items()})


def as_parameters(tensors, keys: Iterable[str]) -> TensorTable:
    """a table in which the columns named in `keys` are fresh leaf Parameters"""
    keys = set(keys)
    return TensorTable({name: torch.nn.Parameter(t.detach(), requires_grad=True) if name in keys else t
                        for name, t in (tensors.items() if hasattr(tensors, "items") else tensors)})


def replace_dict(d: dict, **changes) -> dict:
    return {**d, **changes}


StateTables = Dict[str, TensorTable]  # column name -> the tensor entries of its optimizer state


class ParameterClass:
    def __init__(self, tensors, parameter_groups: Dict[str, Dict],
                 optimizer_state: Optional[Tuple[StateTables, Dict[str, dict]]] = None,
                 optimizer=torch.optim.Optimizer, **optim_kwargs):
        self.tensors = as_parameters(tensors, parameter_groups.keys())
        missing = set(parameter_groups) - set(self.tensors)
        assert not missing, f"parameter groups {sorted(missing)} have no tensor"
        self.optimizer = optimizer([dict(params=[self.tensors[name]], name=name, **options)
                                    for name, options in parameter_groups.items()], **optim_kwargs)
        self.optim_kwargs = optim_kwargs
        if optimizer_state is not None:
            per_row, other = optimizer_state
            for name, entries in per_row.items():
                assert name in self.tensors, f"state parameter {name} not in {list(self.tensors)}"
                rest = {k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in other.get(name, {}).items()}
                self.optimizer.state[self.tensors[name]] = {**dict(entries), **rest}

    # ---- the optimizer's groups
    @property
    def parameter_groups(self) -> Dict[str, dict]:
        return {group["name"]: {k: v for k, v in group.items() if k not in ("params", "name")}
                for group in self.optimizer.param_groups}

    @property
    def learning_rates(self) -> Dict[str, float]:
        return {group["name"]: group["lr"] for group in self.optimizer.param_groups}

    def set_learning_rate(self, **rates: float) -> "ParameterClass":
        for group in self.optimizer.param_groups:
            if group["name"] in rates:
                group["lr"] = rates[group["name"]]
        unknown = set(rates) - set(self.learning_rates)
        assert not unknown, f"no parameter group named {sorted(unknown)}"
        return self

    def update_group(self, name: str, **options) -> None:
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                group.update(options)
                return
        raise ValueError(f"Group {name} not found in optimizer")

    def update_groups(self, **groups) -> Dict[str, float]:
        for name, options in groups.items():
            self.update_group(name, **options)
        return {name: options["lr"] for name, options in groups.items()}

    def zero_grad(self) -> None:
        self.optimizer.zero_grad()

    def step(self, **kwargs) -> None:
        self.optimizer.step(**kwargs)

    # ---- the table
    def keys(self):
        return self.tensors.keys()

    def optimized_keys(self):
        return self.parameter_groups.keys()

    def items(self):
        return self.tensors.items()

    def __getattr__(self, name):
        tensors = self.__dict__.get("tensors")
        if tensors is not None and name in tensors:
            return tensors[name]
        raise AttributeError(name)

    @property
    def batch_size(self) -> torch.Size:
        return self.tensors.batch_size

    @property
    def batch_dims(self) -> int:
        return 1

    def detach(self) -> TensorTable:
        return self.tensors.detach()

    def to_dict(self) -> Dict[str, torch.Tensor]:
        return self.tensors.to_dict()

    # ---- optimizer state, split into what follows the rows and what does not
    def _state_of(self, pick) -> Dict[str, dict]:
        return {name: {k: v for k, v in self.optimizer.state[t].items() if pick(v)}
                for name, t in self.tensors.items() if t in self.optimizer.state}

    def _per_row(self, value) -> bool:
        """state entries that have one row per parameter row (moments, counters); a scalar step count does not"""
        return torch.is_tensor(value) and value.dim() >= 1 and value.shape[0] == int(self.batch_size[0])

    @property
    def tensor_state(self) -> StateTables:
        return {name: TensorTable(entries) for name, entries in self._state_of(self._per_row).items()}

    @property
    def other_state(self) -> Dict[str, dict]:
        return self._state_of(lambda v: not self._per_row(v))

    @property
    def optimizer_state(self) -> Tuple[StateTables, Dict[str, dict]]:
        return self.tensor_state, self.other_state

    def _rebuilt(self, tensors, tensor_state: StateTables) -> "ParameterClass":
        return ParameterClass(tensors, self.parameter_groups, optimizer_state=(tensor_state, self.other_state),
                              optimizer=type(self.optimizer), **self.optim_kwargs)

    def modify_tensors(self, f: Callable[[TensorTable], TensorTable]) -> "ParameterClass":
        """the same parameters after `f` has been applied to the table and to every state table"""
        return self._rebuilt(f(self.tensors), {name: f(table) for name, table in self.tensor_state.items()})

    modify = modify_tensors

    def apply(self, f: Callable[[torch.Tensor], torch.Tensor]) -> "ParameterClass":
        return self.modify_tensors(lambda table: table.apply(f))

    def to(self, device) -> "ParameterClass":
        return self.modify_tensors(lambda table: table.to(device))

    def replace(self, **columns) -> "ParameterClass":
        return self._rebuilt(self.tensors.replace(**columns), self.tensor_state)

    def __getitem__(self, idx: Union[torch.Tensor, str]):
        """a column by name; otherwise the selected rows WITH their optimizer state (pruning)"""
        if isinstance(idx, str):
            return self.tensors[idx]
        return self.modify_tensors(lambda table: table[idx])

    def append_tensors(self, tensors, tensor_state: Optional[StateTables] = None) -> "ParameterClass":
        """rows added at the end (densification); their optimizer state starts at zero unless given"""
        new = TensorTable(tensors)
        assert set(new) == set(self.tensors), f"{sorted(new)} != {sorted(self.tensors)}"
        rows = int(new.batch_size[0])
        state = {}
        for name, table in self.tensor_state.items():
            extra = table.new_zeros(rows) if tensor_state is None else TensorTable(tensor_state[name])
            assert int(extra.batch_size[0]) == rows, f"{name}: state for {extra.batch_size[0]} rows, {rows} appended"
            state[name] = TensorTable.concat(table, extra)
        return self._rebuilt(TensorTable.concat(self.tensors.detach(), new.detach()), state)

    def append(self, params: "ParameterClass") -> "ParameterClass":
        return self.append_tensors(params.tensors)

    # ---- persistence
    def state_dict(self) -> dict:
        per_row, other = self.optimizer_state
        return dict(tensors=self.tensors.detach().to_dict(),
                    optimizer=({name: table.to_dict() for name, table in per_row.items()}, other),
                    parameter_groups=self.parameter_groups)

    @staticmethod
    def from_state_dict(state: dict, optimizer=torch.optim.Adam, **optim_kwargs) -> "ParameterClass":
        per_row, other = state["optimizer"]
        return ParameterClass(TensorTable(state["tensors"]), parameter_groups=state["parameter_groups"],
                              optimizer_state=({name: TensorTable(t) for name, t in per_row.items()}, other),
                              optimizer=optimizer, **optim_kwargs)
