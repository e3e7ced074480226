% Drop-in for src/primal/ManiSDP_unittrace.m:  Min <C,X> s.t. A(X) = b, X >= 0, tr(X) = 1.
% Same signature / options / outputs (Y is n x p here, as in the reference); trustregions() ->
% manisdp_mex('rtr', ...) on the sphere manifold.
function [X, obj, data] = ManiSDP_unittrace(At, b, c, K, options)

n = K.s;
if ~isfield(options,'p0'); options.p0 = 1; end
if ~isfield(options,'AL_maxiter'); options.AL_maxiter = 1000; end
if ~isfield(options,'gama'); options.gama = 2; end
if ~isfield(options,'sigma0'); options.sigma0 = 1e1; end
if ~isfield(options,'sigma_min'); options.sigma_min = 1e2; end
if ~isfield(options,'sigma_max'); options.sigma_max = 1e7; end
if ~isfield(options,'tol'); options.tol = 1e-8; end
if ~isfield(options,'theta'); options.theta = 1e-2; end
if ~isfield(options,'delta'); options.delta = 8; end
if ~isfield(options,'alpha'); options.alpha = 0.05; end
if ~isfield(options,'tolgradnorm'); options.tolgradnorm = 1e-8; end
if ~isfield(options,'TR_maxinner'); options.TR_maxinner = 40; end
if ~isfield(options,'TR_maxiter'); options.TR_maxiter = 3; end
if ~isfield(options,'tau1'); options.tau1 = 1e-5; end
if ~isfield(options,'tau2'); options.tau2 = 1e-4; end
if ~isfield(options,'line_search'); options.line_search = 1; end

fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', n, size(b,1));

A = At';
h = manisdp_mex('create_unittrace', At, b, c, n);
cleanup = onCleanup(@() manisdp_mex('destroy', h));
p = options.p0;
sigma = options.sigma0;
gama = options.gama;
y = zeros(length(b),1);
normb = 1 + norm(b);
if isfield(options, 'Y0')
    Y = options.Y0;
else
    Y = randn(n, p);  Y = Y/norm(Y, 'fro');        % spherefactory rand
end
U = [];
opts.maxinner = options.TR_maxinner;
opts.maxiter = options.TR_maxiter;
opts.tolgradnorm = options.tolgradnorm;
opts.unittrace = 1;

data.status = 0;
timespend = tic;
for iter = 1:options.AL_maxiter
    manisdp_mex('set_multipliers', h, y, sigma);
    if ~isempty(U)
        Y = line_search(Y, U);
    end
    [Y, info] = manisdp_mex('rtr', h, Y, opts);
    gradnorm = info.gradnorm;
    X = Y*Y';
    x = X(:);
    obj = c'*x;
    Axb = A*x - b;
    pinf = norm(Axb)/normb;
    y = y - sigma*Axb;
    eS = reshape(c - At*y, n, n);
    z = sum(eS.*X, 'all');
    S = eS - z*eye(n);
    [vS, dS] = eig(full(S), 'vector');
    dinf = max(0, -dS(1))/(1+dS(end));
    by = b'*y + z;
    gap = abs(obj-by)/(abs(by)+abs(obj)+1);
    [V, D, ~] = svd(Y, 'econ');
    if size(D, 2) > 1
        e = diag(D);
    else
        e = D(1);
    end
    r = sum(e >= options.theta*e(1));
    fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs\n', ...
             iter,    obj,       gap,       pinf,       dinf,       gradnorm,    r,    p,    sigma,   toc(timespend));
    eta = max([pinf, gap, dinf]);
    if eta < options.tol
        fprintf('Optimality is reached!\n');
        break;
    end
    if mod(iter, 20) == 0
        if iter > 50 && gap > gap0 && pinf > pinf0 && dinf > dinf0
            data.status = 2;
            fprintf('Slow progress!\n');
            break;
        else
            gap0 = gap; pinf0 = pinf; dinf0 = dinf;
        end
    end
    if r <= p - 1
        Y = V(:,1:r)*diag(e(1:r));
        p = r;
    end
    nne = min(sum(dS < 0), options.delta);
    if options.line_search == 1
       U = [zeros(n, p) vS(:,1:nne)];
    end
    p = p + nne;
    if options.line_search == 1
        Y = [Y zeros(n, nne)];
    else
        Y = [Y options.alpha*vS(:,1:nne)];
        Y = Y/norm(Y, 'fro');
    end
    if pinf < options.tau1*gradnorm
          sigma = max(sigma/gama, options.sigma_min);
    elseif pinf > options.tau2*gradnorm
          sigma = min(sigma*gama, options.sigma_max);
    end
end
data.X = X; data.y = y; data.S = S; data.z = z; data.gap = gap; data.pinf = pinf; data.dinf = dinf;
data.gradnorm = gradnorm; data.time = toc(timespend);
if data.status == 0 && eta > options.tol
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(timespend));

    function nY = line_search(Y, U)
         alpha = 1;
         cost0 = manisdp_mex('linesearch_cost', h, Y, U, 0);
         i = 1;
         nY = Y + alpha*U;  nY = nY/norm(nY, 'fro');
         while i <= 15 && manisdp_mex('linesearch_cost', h, Y, U, alpha) - cost0 > -1e-3
              alpha = 0.8*alpha;
              nY = Y + alpha*U;  nY = nY/norm(nY, 'fro');
              i = i + 1;
         end
    end
end
