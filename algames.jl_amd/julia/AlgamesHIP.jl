# AlgamesHIP.jl -- Julia host shim over the C ABI of libalgames_hip.so (include/algames_hip.h).
#
# UNEXECUTED IN THIS ENVIRONMENT: neither the build container nor the GPU box has a `julia` binary
# (SURVEY.md section 0), so this file is the binding a maintainer of Algames.jl would add; every call it makes
# is exercised through the identical ctypes binding (algames.jl_amd/_abi.py) by the test-suite.
#
# It keeps the reference API surface for the hot path: `GameProblem` / `Options` are the reference's own
# types; `newton_solve!(probs::Vector{<:GameProblem})` solves a batch of structurally identical problems
# on one MI355X and writes the results back into each `prob.pdtraj`, the constraint multipliers and
# `prob.stats` -- replacing src/problem/solver_methods.jl:5-65 for that batch.
module AlgamesHIP

using Algames
using StaticArrays
import Algames: newton_solve!

const LIB = get(ENV, "ALGAMES_HIP_LIB", joinpath(@__DIR__, "..", "lib", "libalgames_hip.so"))

struct AlgDesc
    model::Int32; p::Int32; d::Int32; N::Int32
    dt::Float64
    batch::Int32; device::Int32
end

struct AlgOptions                      # POD mirror of include/algames_hip.h: alg_options
    amplitude_init::Float64; shift::Int32; regularize::Int32
    reg_0::Float64; alpha_decrease::Float64; beta::Float64
    ls_iter::Int32; dual_reset::Int32; delta_min::Float64
    rho_0::Float64; rho_increase::Float64; rho_max::Float64; lambda_max::Float64; alpha_dual::Float64
    alphax_dual::NTuple{10,Float64}
    eps_dyn::Float64; eps_sta::Float64; eps_con::Float64; eps_opt::Float64
    outer_iter::Int32; inner_iter::Int32
    seed::Int64
end

struct AlgRecord
    outer::Int32; ls_j::Int32
    alpha::Float64; res::Float64; delta::Float64
    dyn_vio::Float64; con_vio::Float64; sta_vio::Float64; opt_vio::Float64
end

struct AlgGameStats
    status::Int32; outer_iters::Int32; newton_iters::Int32; records::Int32; converged::Int32; ls_failures::Int32
    last::AlgRecord
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:alg_last_error, LIB), Cstring, ())))

model_id(::DoubleIntegratorGame) = Int32(0)
model_id(::UnicycleGame) = Int32(1)
model_id(::BicycleGame) = Int32(2)

function abi_options(o::Options)
    ax = ntuple(i -> i <= length(o.αx_dual) ? Float64(o.αx_dual[i]) : 1.0, 10)
    AlgOptions(o.amplitude_init, min(o.shift, 2^30), o.regularize, o.reg_0, o.α_decrease, o.β, o.ls_iter,
               o.dual_reset, o.Δ_min, o.ρ_0, o.ρ_increase, o.ρ_max, o.λ_max, o.α_dual, ax,
               o.ϵ_dyn, o.ϵ_sta, o.ϵ_con, o.ϵ_opt, o.outer_iter, o.inner_iter, o.seed)
end

"""
    newton_solve!(probs::Vector{<:GameProblem}; device=0, game_id0=0)

Batched drop-in for `newton_solve!(prob)` (src/problem/solver_methods.jl:5-65).  All problems must share
model, N, dt, options and constraint structure (collision avoidance radii, control / state bounds, walls, circles,
collision cost);
they may differ in x0 and in the LQR data.
"""
function newton_solve!(probs::Vector{<:GameProblem}; device::Integer=0, game_id0::Integer=0)
    prob = probs[1]; ps = prob.probsize; B = length(probs)
    N, n, m, p = ps.N, ps.n, ps.m, ps.p
    ni, mi = ps.ni[1], ps.mi[1]
    d = prob.model isa DoubleIntegratorGame ? mi : 2
    dt = prob.pdtraj.pr[1].dt
    desc = Ref(AlgDesc(model_id(prob.model), p, d, N, dt, B, device))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:alg_create, LIB), Cint, (Ref{AlgDesc}, Ref{Ptr{Cvoid}}), desc, h))
    try
        check(ccall((:alg_set_options, LIB), Cint, (Ptr{Cvoid}, Ref{AlgOptions}), h[], Ref(abi_options(prob.opts))))
        # x0: B x n, game-major (Julia is column-major: build n x B)
        x0 = hcat([Vector(pr.x0) for pr in probs]...)
        check(ccall((:alg_set_x0, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], x0))
        # LQR data from game_obj.obj[i][1] (LQRCost: Q diagonal, q = -Q xf), per game: (ni x p x B) in memory order
        Qd = zeros(ni, p, B); Rd = zeros(mi, p, B); xf = zeros(ni, p, B); uf = zeros(mi, p, B)
        for (g, pr) in enumerate(probs), i in 1:p
            c = pr.game_obj.obj[i][1].cost[1]
            Qfull = diag(c.Q); Rfull = diag(c.R)
            Qd[:, i, g] = Qfull[ps.pz[i]]; Rd[:, i, g] = Rfull[ps.pu[i]]
            xf[:, i, g] = -(c.q ./ map(x -> x == 0 ? 1.0 : x, Qfull))[ps.pz[i]]
            uf[:, i, g] = -(c.r ./ map(x -> x == 0 ? 1.0 : x, Rfull))[ps.pu[i]]
        end
        check(ccall((:alg_set_lqr, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32),
                    h[], Qd, Rd, xf, uf, 1))
        # collision costs: game_obj.obj[i][2:end] are CollisionCost objectives (objective.jl:84-100)
        if length(prob.game_obj.obj[1]) > 1
            rad = [prob.game_obj.obj[i][2].cost[1].r for i in 1:p]
            mu = [prob.game_obj.obj[i][2].cost[1].μ for i in 1:p]
            check(ccall((:alg_add_collision_cost, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], rad, mu))
        end
        # state constraints of player i, in the order they were added (constraints_methods.jl): dispatch on the type
        colcons = [cv.con for cv in prob.game_con.state_conval[1] if cv.con isa Algames.TrajectoryOptimization.CollisionConstraint]
        if p > 1 && !isempty(colcons)
            # collision avoidance: con.radius = r_i + r_j (constraints_methods.jl:27-29) -> per-player radii
            col2 = [cv.con for cv in prob.game_con.state_conval[2] if cv.con isa Algames.TrajectoryOptimization.CollisionConstraint]
            R12 = colcons[1].radius                                     # r_1 + r_2
            R1p = p > 2 ? colcons[2].radius : R12                       # r_1 + r_3
            R2p = p > 2 ? col2[2].radius : R12                          # r_2 + r_3
            r1 = p > 2 ? (R12 + R1p - R2p) / 2 : R12 / 2
            radius = [i == 1 ? r1 : colcons[i-1].radius - r1 for i in 1:p]
            # add_spherical_collision_avoidance! builds the constraint on pz[i][1:3] (three indices) instead of px[i] (two)
            spherical = length(colcons[1].x1) == 3
            check(spherical ? ccall((:alg_add_spherical_collision_avoidance, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], radius) :
                              ccall((:alg_add_collision_avoidance, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], radius))
        end
        if prob.model isa BicycleGame
            check(ccall((:alg_set_bicycle, LIB), Cint, (Ptr{Cvoid}, Float64, Float64), h[], prob.model.lf, prob.model.lr))
        end
        for i in 1:p, cv in prob.game_con.state_conval[i]
            con = cv.con
            if con isa Algames.StateBoundConstraint                      # add_state_bound!(game_con, i, x_max, x_min)
                check(ccall((:alg_add_state_bound, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), h[], i - 1,
                            Vector{Float64}(con.x_max), Vector{Float64}(con.x_min)))
            elseif i == 1 && con isa Algames.WallConstraint              # add_wall_constraint!(game_con, walls): same set for every player
                check(ccall((:alg_add_wall_constraint, LIB), Cint,
                            (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], length(con),
                            Vector{Float64}(con.x1), Vector{Float64}(con.y1), Vector{Float64}(con.x2), Vector{Float64}(con.y2),
                            Vector{Float64}(con.xv), Vector{Float64}(con.yv)))
            elseif i == 1 && con isa Algames.TrajectoryOptimization.CircleConstraint   # add_circle_constraint!(game_con, xc, yc, radius)
                check(ccall((:alg_add_circle_constraint, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], length(con),
                            Vector{Float64}(con.x), Vector{Float64}(con.y), Vector{Float64}(con.radius)))
            elseif i == 1 && con isa Algames.Wall3DConstraint            # add_wall_constraint!(game_con, walls::Vector{Wall3D}); n_wall x 3 row-major
                pts(a, b, c) = Vector{Float64}(vec(permutedims(hcat(Vector(a), Vector(b), Vector(c)))))
                check(ccall((:alg_add_wall3d_constraint, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], length(con),
                            pts(con.x1, con.y1, con.z1), pts(con.x2, con.y2, con.z2), pts(con.x3, con.y3, con.z3), pts(con.xv, con.yv, con.zv)))
            elseif i == 1 && con isa Algames.CylinderConstraint          # add_wall_constraint!(game_con, walls::Vector{CylinderWall})
                pts(a, b, c) = Vector{Float64}(vec(permutedims(hcat(Vector(a), Vector(b), Vector(c)))))
                axis = Int32[s == :x ? 0 : s == :y ? 1 : 2 for s in con.v]
                check(ccall((:alg_add_cylinder_constraint, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), h[], length(con),
                            pts(con.p1, con.p2, con.p3), axis, Vector{Float64}(con.l), Vector{Float64}(con.r)))
            end
        end
        if !isempty(prob.game_con.control_conval)
            con = prob.game_con.control_conval[1].con
            check(ccall((:alg_add_control_bound, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[],
                        Vector(con.u_max), Vector(con.u_min)))
        end
        stats = Vector{AlgGameStats}(undef, B)
        check(ccall((:alg_newton_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{AlgGameStats}), h[], 1, game_id0, stats))
        # write back: pdtraj (x_1 | horizontal-order vector, primal_dual_traj.jl:46-75), multipliers, stats
        S = ps.S
        z = zeros(n + S, B)
        check(ccall((:alg_get_traj, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), h[], 0, z))
        for (g, pr) in enumerate(probs)
            Algames.set_traj!(pr.core, pr.pdtraj, view(z, n+1:n+S, g))
            Algames.RobotDynamics.set_state!(pr.pdtraj.pr[1], SVector{n}(z[1:n, g]))
        end
        # (constraint multipliers lambda / mu: alg_get_con_duals, layout documented in include/algames_hip.h;
        #  Statistics history: alg_get_history -> record!(stats, ...) with the stored scalars)
        return stats
    finally
        ccall((:alg_destroy, LIB), Cvoid, (Ptr{Cvoid},), h[])
    end
end

# The other drop-ins bind the same way on a handle prepared as above (same setup calls, then instead of alg_newton_solve):
#
#   ibr_newton_solve!(prob; ibr_opts)   (src/problem/solver_methods.jl:133-169)
#       ordering = Int32.(ibr_opts.ordering .- 1)
#       check(ccall((:alg_ibr_newton_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Int32, Ptr{Int32}, Float64, Ptr{AlgGameStats}),
#                   h[], 1, game_id0, ibr_opts.ibr_iter, ordering, ibr_opts.Δ_min, stats))
#   ibr_newton_solve!(prob, i)          (solver_methods.jl:171-228)
#       check(ccall((:alg_ibr_solve_player, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{AlgGameStats}), h[], i - 1, stats))
#   receding-horizon loop of BASELINE config 5 (opts.shift / opts.dual_reset warm starts; `steps` MPC steps per game, one launch)
#       states = zeros(n, B, steps + 1)
#       check(ccall((:alg_mpc_solve, LIB), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{Float64}), h[], steps, game_id0, states))
#       check(ccall((:alg_mpc_totals, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int32), h[], iters, converged, 0))
#   step-wise entry points (residual!, residual_jacobian!, inner_iteration, line_search, dual / penalty update):
#       alg_residual, alg_residual_jacobian, alg_newton_step, alg_line_search, alg_dual_penalty_update (include/algames_hip.h)

end # module
