"""wrap_agentid (reference: learning_strategies/evolution/utils.py:4-8): every agent of a team gets its own
copy of the same network (shared weights, private recurrent state)."""
import copy


def wrap_agentid(agent_ids, network):
    return {aid: copy.deepcopy(network) for aid in agent_ids}
