"""Runner of the GAN (JS) imitation policy (reference gan/runner.py:13-342): build the models and
optimisers, load + window the expert trajectories, epoch loop {dynamics, critic, cost}, save.

Differences from the reference, all forced by what ships: the state / action sizes come from the
trajectory file (no dm_control here), `dataset_path` names that file, PRNG keys are NumPy generators,
and the environment-driven parts (dynamics_trainer's rollouts in MuJoCo, the final reward average,
the video) run only when an `env` object with the dm_control protocol is passed in."""

import numpy as np

from gan_mpc_amd import data_buffers, data_loader, runner_common, utils
from gan_mpc_amd.gan import critic_trainer, js_policy
from gan_mpc_amd.norm import cost_trainer
from gan_mpc_amd.policy import eval as eval_policy_mod

get_optimizer = runner_common.get_optimizer
get_normalizer = runner_common.get_normalizer


def get_policy(config, x_size, u_size, expert=None):
    cost, _ = utils.get_cost_model(config)
    dynamics, _ = utils.get_dynamics_model(config, x_size)
    expert = expert or utils.get_expert_model(config, x_size, u_size)
    critic, _ = utils.get_critic_model(config)
    train_policy = js_policy.JS_MPC(config=config, cost_model=cost, dynamics_model=dynamics,
                                    expert_model=expert, critic_model=critic)
    eval_policy = eval_policy_mod.EvalMPC(config=config, cost_model=cost, dynamics_model=dynamics,
                                          expert_model=expert)
    return train_policy, eval_policy, config.mpc


def get_params(policy, config, x_size, u_size):
    return runner_common.get_params(policy, config, x_size, u_size, with_critic=True)


def train(config, env, policy_args, cost_opt_args, dynamics_opt_args, critic_opt_args, buffers,
          cost_dataset, dynamics_dataset, key):
    """reference gan/runner.py:84-209"""
    train_policy, eval_policy, params = policy_args
    cost_opt, cost_opt_state = cost_opt_args
    dynamics_opt, dynamics_opt_state = dynamics_opt_args
    critic_opt, critic_opt_state = critic_opt_args
    num_epochs = config.mpc.train.num_epochs
    print_after_n_epochs = config.mpc.train.print_after_n_epochs
    cost_config = config.mpc.train.cost
    critic_config = config.mpc.train.critic
    cost_train_losses, cost_test_losses = [], []
    dynamics_train_losses, dynamics_test_losses = [0.0], [0.0]     # the reference's defaults
    dynamics_env_rewards = [[0.0]]
    critic_train_losses, critic_test_losses = [0.0], [0.0]
    dynamics_exe_time = 0.0
    params = train_policy.to_device_params(params)
    for ep in range(1, num_epochs + 1):
        key, (subkey1, subkey2, subkey3) = runner_common.split_keys(key, 3)
        if env is not None or dynamics_dataset is not None:
            from gan_mpc_amd.norm import dynamics_trainer
            dynamics_config = config.mpc.train.dynamics
            (params, dynamics_opt_state, buffers, ep_rewards, ep_dyn_train, ep_dyn_test,
             dynamics_exe_time) = dynamics_trainer.train(
                env=env, train_args=(train_policy, eval_policy, dynamics_opt),
                opt_state=dynamics_opt_state, params=params, dataset=dynamics_dataset,
                buffers=buffers, num_episodes=dynamics_config.num_episodes,
                max_interactions_per_episode=dynamics_config.max_interactions_per_episode,
                num_updates=dynamics_config.num_updates, batch_size=dynamics_config.batch_size,
                discount_factor=dynamics_config.discount_factor,
                teacher_forcing_factor=dynamics_config.teacher_forcing_factor, key=subkey1, id=ep)
            dynamics_env_rewards.extend(ep_rewards)
            dynamics_train_losses.extend(ep_dyn_train)
            dynamics_test_losses.extend(ep_dyn_test)

        (params, critic_opt_state, ep_critic_train, ep_critic_test,
         critic_exe_time) = critic_trainer.train(
            train_args=(train_policy, critic_opt), opt_state=critic_opt_state, params=params,
            true_dataset=cost_dataset, num_updates=critic_config.num_updates,
            batch_size=critic_config.batch_size, key=subkey2, id=ep)

        (params, cost_opt_state, ep_cost_train, ep_cost_test, cost_exe_time) = cost_trainer.train(
            train_args=(train_policy, cost_opt), opt_state=cost_opt_state, params=params,
            dataset=cost_dataset, num_updates=cost_config.num_updates,
            batch_size=cost_config.batch_size, polyak_factor=cost_config.polyak_factor,
            key=subkey3, id=ep)

        critic_train_losses.extend(ep_critic_train)
        critic_test_losses.extend(ep_critic_test)
        cost_train_losses.extend(ep_cost_train)
        cost_test_losses.extend(ep_cost_test)

        if (ep % print_after_n_epochs) == 0:
            print("-----------------------------")
            print(f"epoch: {ep} env_reward: {sum(dynamics_env_rewards[-1]):.2f}")
            print(f"dyna_exe_time: {dynamics_exe_time:.2f} mins, "
                  f"dyna_train_loss: {dynamics_train_losses[-1]:.5f}, "
                  f"dyna_test_loss: {dynamics_test_losses[-1]:.5f}")
            print(f"critic_exe_time: {critic_exe_time:.2f} mins, "
                  f"critic_train_loss: {critic_train_losses[-1]:.5f}, "
                  f"critic_test_loss: {critic_test_losses[-1]:.5f}")
            print(f"cost_exe_time: {cost_exe_time:.2f} mins, "
                  f"cost_train_loss: {cost_train_losses[-1]:.5f}, "
                  f"cost_test_loss: {cost_test_losses[-1]:.5f}")

    return (params, (dynamics_env_rewards, dynamics_train_losses, dynamics_test_losses),
            (critic_train_losses, critic_test_losses), (cost_train_losses, cost_test_losses))


def run(config_path, dataset_path=None, env=None, expert=None, save_dir=None):
    """reference gan/runner.py:212-337"""
    config = utils.get_config(config_path)
    key = np.random.default_rng(config.seed)

    normalizer = get_normalizer(config.mpc.normalizer)
    dataloader = data_loader.DataLoader(config=config, normalizer=normalizer).init(path=dataset_path)
    x_size = dataloader.expert_trajectories["states"].shape[-1]
    u_size = dataloader.expert_trajectories["actions"].shape[-1]

    train_policy, eval_policy, policy_config = get_policy(config, x_size, u_size, expert=expert)
    params = train_policy.to_device_params(get_params(train_policy, config, x_size, u_size))

    tr = config.mpc.train
    cost_opt_args = get_optimizer(params, tr.cost.no_grads, tr.cost.learning_rate)
    dynamics_opt_args = get_optimizer(params, tr.dynamics.no_grads, tr.dynamics.learning_rate)
    critic_opt_args = get_optimizer(params, tr.critic.no_grads, tr.critic.learning_rate)

    key, (subkey1, subkey2) = runner_common.split_keys(key, 2)
    cost_dataset = dataloader.get_cost_dataset(subkey1)
    dynamics_dataset = dataloader.get_dynamics_dataset(subkey2)

    replay_buffer = data_buffers.ReplayBuffer(horizon=config.mpc.horizon,
                                              q_maxlen=tr.dynamics.replay_buffer_size,
                                              normalizer=dataloader.normalizer)
    buffer = data_buffers.Buffer(maxlen=config.mpc.horizon, normalizer=dataloader.normalizer)

    params, dynamics_out_args, critic_out_args, cost_out_args = train(
        config=config, env=env, policy_args=(train_policy, eval_policy, params),
        cost_opt_args=cost_opt_args, dynamics_opt_args=dynamics_opt_args,
        critic_opt_args=critic_opt_args, buffers=(replay_buffer, buffer),
        cost_dataset=cost_dataset, dynamics_dataset=dynamics_dataset, key=key)

    dynamics_env_rewards, dynamics_train_losses, dynamics_test_losses = dynamics_out_args
    critic_train_losses, critic_test_losses = critic_out_args
    cost_train_losses, cost_test_losses = cost_out_args

    avg_reward = 0.0
    if env is not None:
        from gan_mpc_amd.norm import dynamics_trainer
        avg_reward = dynamics_trainer.avg_run_policy(
            env=env, policy_fn=eval_policy.get_optimal_action, params=params, buffer=buffer,
            max_interactions=config.mpc.evaluate.max_interactions,
            num_runs=config.mpc.evaluate.num_runs_for_avg)

    save_config = {
        "seed": config.seed,
        "env": config.env.to_dict(),
        "loss": {
            "dynamics": {"train_loss": round(dynamics_train_losses[-1], 5),
                         "test_loss": round(dynamics_test_losses[-1], 5)},
            "cost": {"train_loss": round(cost_train_losses[-1], 5),
                     "test_loss": round(cost_test_losses[-1], 5)},
            "critic": {"train_loss": round(critic_train_losses[-1], 5),
                       "test_loss": round(critic_test_losses[-1], 5)},
        },
        "reward": round(float(avg_reward), 2),
        "policy": policy_config.to_dict(),
    }
    env_type, env_name = config.env.type, config.env.expert.name
    dir_path = save_dir or f"trained_models/imitator/{env_type}/{env_name}/gan/"
    abs_dir_path = utils.save_all_args(
        dir_path, params, save_config,
        (dynamics_env_rewards, "dynamics_env_rewards.json"),
        (dynamics_train_losses, "dynamics_train_losses.json"),
        (dynamics_test_losses, "dynamics_test_losses.json"),
        (critic_train_losses, "critic_train_losses.json"),
        (critic_test_losses, "critic_test_losses.json"),
        (cost_train_losses, "cost_train_losses.json"),
        (cost_test_losses, "cost_test_losses.json"))
    return abs_dir_path


if __name__ == "__main__":
    import sys
    run(config_path=sys.argv[1] if len(sys.argv) > 1 else "config/gan_hyperparameters.yaml",
        dataset_path=sys.argv[2] if len(sys.argv) > 2 else None)
