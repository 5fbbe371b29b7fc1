"""Reference class name + constructor kwargs -> TabularModel.

Mirrors the constructor signatures of the reference's MDP classes (BaseMDP.__init__,
colosseum/mdp/base.py:327-343, plus each family's own keywords) so that a gin/yaml
configuration written for the reference can be fed here unchanged."""
from typing import Any, Dict

from .builder import TabularModel, build_model
from .families import Custom, DeepSea, FrozenLake, MiniGridEmpty, MiniGridRooms, RiverSwim, SimpleGrid, Taxi

_BASE_KEYS = ("randomize_actions", "p_lazy", "p_rand", "rewards_range")
_IGNORED = ("emission_map", "emission_map_kwargs", "noise", "noise_kwargs", "instantiate_mdp",
            "force_sparse_transition", "exclude_horizon_from_parameters")

FAMILIES = {
    "DeepSea": DeepSea,
    "FrozenLake": FrozenLake,
    "MiniGridEmpty": MiniGridEmpty,
    "MiniGridRooms": MiniGridRooms,
    "RiverSwim": RiverSwim,
    "SimpleGrid": SimpleGrid,
    "Taxi": Taxi,
    "Custom": Custom,
}


def split_class_name(cls_name: str):
    for suffix, episodic in (("Episodic", True), ("Continuous", False)):
        if cls_name.endswith(suffix):
            fam = cls_name[: -len(suffix)]
            if fam in FAMILIES:
                return fam, episodic
    raise KeyError(f"{cls_name!r} is not one of the supported MDP classes "
                   f"({', '.join(f + s for f in FAMILIES for s in ('Episodic', 'Continuous'))})")


def make_model(cls_name: str, **kwargs: Any) -> TabularModel:
    fam_name, episodic = split_class_name(cls_name)
    kw: Dict[str, Any] = dict(kwargs)
    seed = kw.pop("seed")
    for k in _IGNORED:
        v = kw.pop(k, None)
        if k in ("emission_map", "noise") and v is not None and getattr(v, "__name__", "") != "Tabular":
            raise NotImplementedError("only the tabular emission map is in scope (SURVEY.md section 2, rows 13-14)")
    base = {k: kw.pop(k) for k in _BASE_KEYS if k in kw}
    H = kw.pop("H", None)
    if fam_name == "DeepSea" and episodic:
        if "size" not in kw:
            raise NotImplementedError("The 'size' parameter should be given as a keyword parameter.")
        H = kw["size"]  # deep_sea/finite_horizon.py:28-36
    if fam_name == "FrozenLake":
        family = FrozenLake(seed=seed, **kw)
    elif fam_name in ("RiverSwim", "Taxi"):  # their reward defaults / checks depend on the setting
        family = FAMILIES[fam_name](episodic=episodic, **kw)
    else:
        family = FAMILIES[fam_name](**kw)
    if getattr(family, "force_randomize_actions", None) is not None:
        base["randomize_actions"] = family.force_randomize_actions  # taxi/base.py:488-490
    model = build_model(family, seed, episodic, H=H, **base)
    model.extra["family"] = family
    model.extra["cls_name"] = cls_name
    model.extra["kwargs"] = dict(kwargs)
    return model
