"""Entry points of the L2 imitation policy with the reference's names (reference norm/runner.py).
The program itself is gan_mpc_amd/runner_common.py, shared with gan/runner.py."""

from gan_mpc_amd import runner_common, utils
from gan_mpc_amd.norm import l2_policy
from gan_mpc_amd.policy import eval as eval_policy_mod

get_optimizer = runner_common.get_optimizer
get_normalizer = runner_common.get_normalizer


def get_policy(config, x_size, u_size, expert=None):
    models = dict(cost_model=utils.get_cost_model(config)[0],
                  dynamics_model=utils.get_dynamics_model(config, x_size)[0],
                  expert_model=expert or utils.get_expert_model(config, x_size, u_size))
    return (l2_policy.L2MPC(config=config, **models), eval_policy_mod.EvalMPC(config=config, **models),
            config.mpc)


def get_params(policy, config, x_size, u_size):
    return runner_common.get_params(policy, config, x_size, u_size, with_critic=False)


def train(config, env, policy_args, cost_opt_args, dynamics_opt_args, buffers, cost_dataset,
          dynamics_dataset, key):
    """-> (params, (env rewards, dynamics train, dynamics test), (cost train, test))"""
    train_policy, eval_policy, params = policy_args
    opts = {"cost": cost_opt_args, "dynamics": dynamics_opt_args}
    params, rewards, h = runner_common.train_loop(config, env, train_policy, eval_policy, params, opts,
                                                  buffers, cost_dataset, dynamics_dataset, key, False)
    return (params, (rewards, h["dynamics"].train, h["dynamics"].test), (h["cost"].train, h["cost"].test))


def run(config_path, dataset_path=None, env=None, expert=None, save_dir=None):
    return runner_common.run("l2", get_policy, config_path, dataset_path, env, expert, save_dir)


if __name__ == "__main__":
    import sys
    run(config_path=sys.argv[1] if len(sys.argv) > 1 else "config/l2_hyperparameters.yaml",
        dataset_path=sys.argv[2] if len(sys.argv) > 2 else None)
