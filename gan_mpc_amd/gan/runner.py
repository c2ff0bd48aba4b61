"""Entry points of the GAN (JS) imitation policy with the reference's names (reference
gan/runner.py: get_policy, get_params, get_optimizer, get_normalizer, train, run).  The program itself
is gan_mpc_amd/runner_common.py, shared with norm/runner.py."""

from gan_mpc_amd import runner_common, utils
from gan_mpc_amd.gan import js_policy
from gan_mpc_amd.policy import eval as eval_policy_mod

get_optimizer = runner_common.get_optimizer
get_normalizer = runner_common.get_normalizer


def get_policy(config, x_size, u_size, expert=None):
    """Training policy (with the critic) and evaluation policy over the same models."""
    models = dict(cost_model=utils.get_cost_model(config)[0],
                  dynamics_model=utils.get_dynamics_model(config, x_size)[0],
                  expert_model=expert or utils.get_expert_model(config, x_size, u_size))
    train_policy = js_policy.JS_MPC(config=config, critic_model=utils.get_critic_model(config)[0], **models)
    return train_policy, eval_policy_mod.EvalMPC(config=config, **models), config.mpc


def get_params(policy, config, x_size, u_size):
    return runner_common.get_params(policy, config, x_size, u_size, with_critic=True)


def train(config, env, policy_args, cost_opt_args, dynamics_opt_args, critic_opt_args, buffers,
          cost_dataset, dynamics_dataset, key):
    """-> (params, (env rewards, dynamics train, dynamics test), (critic train, test), (cost train, test))"""
    train_policy, eval_policy, params = policy_args
    opts = {"cost": cost_opt_args, "dynamics": dynamics_opt_args, "critic": critic_opt_args}
    params, rewards, h = runner_common.train_loop(config, env, train_policy, eval_policy, params, opts,
                                                  buffers, cost_dataset, dynamics_dataset, key, True)
    return (params, (rewards, h["dynamics"].train, h["dynamics"].test),
            (h["critic"].train, h["critic"].test), (h["cost"].train, h["cost"].test))


def run(config_path, dataset_path=None, env=None, expert=None, save_dir=None):
    return runner_common.run("gan", get_policy, config_path, dataset_path, env, expert, save_dir)


if __name__ == "__main__":
    import sys
    run(config_path=sys.argv[1] if len(sys.argv) > 1 else "config/gan_hyperparameters.yaml",
        dataset_path=sys.argv[2] if len(sys.argv) > 2 else None)
